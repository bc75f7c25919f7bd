// be_kernels.h — argument blocks and launchers of the back-end kernels (product code, not part of the ABI).
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include "be_factor_dev.h"

#define IMU_OUT_STRIDE 936

// Landmark-sharded window (north_star: "all-reduce of the reduced camera-pose Hessian"): rank r of `world` owns the landmarks [lo, hi)
// (contiguous, cap = ceil(nlm / world) each).  Every rank evaluates and Schur-reduces only its own landmarks; IMU factors and the prior are evaluated
// on every rank (identical bits, no exchange).  Round 5: the exchange no longer depends on the number of landmarks.  What the trust-region step needs of
// the landmarks it does not own are SUMS — norms of the Gauss-Newton / gradient / dogleg steps, the model cost change — and every one of them is a
// quadratic form in the pose step of the same iteration: with v = s_p y_p (the pose part of the Gauss-Newton step), a landmark's step is
//     dl = (cg alpha - cn rho) g + cn rho (w . v),   alpha = s^2 / d^2,  rho = 1 / (h + mu / alpha)
// so that  sum dl^2, sum h dl^2, sum dl g, sum dl (w . delta_p), ...  are  scalar + vector . v + v^T matrix v  with weights known at reduce time.
// One exchange vector per rank and linearisation, BE_XS_LEN doubles (99 KB) whatever the window holds:
//   M(k), k = 0..4   66 lower block pairs (fi >= fj) x 36:  sum rho w w^T | sum D (direct pose-pose terms) | sum rho^2/alpha w w^T | sum h rho^2 w w^T | sum rho^2 w w^T
//   V(k), k = 0..7   66 each:  sum g_p | sum rho g w | sum rho^2/alpha g w | sum alpha g w | sum h alpha rho g w | sum h rho^2 g w | sum alpha rho g w | sum rho^2 g w
//   A                16 scalars: sum alpha g^2, rho^2/alpha g^2, rho g^2, h alpha^2 g^2, h alpha rho g^2, h rho^2 g^2, alpha^2 g^2, alpha rho g^2, rho^2 g^2, x^2, cost; max |g|
// all-gathered and summed IN RANK ORDER by be_shard_finalize_kernel (identical bits on every rank -> identical decisions -> the same number of exchanges);
// every rank back-substitutes, norms and moves its OWN landmarks only (be_solve_shard_kernel); the candidate costs travel as one partial sum per rank; the
// inverse depths are gathered once, behind the last iteration slot.
#define BE_XS_MSZ (66 * 36)
#define BE_XS_M(k) ((k) * BE_XS_MSZ)
#define BE_XS_V(k) (5 * BE_XS_MSZ + (k) * 66)
#define BE_XS_A (5 * BE_XS_MSZ + 8 * 66)
#define BE_XS_LEN (BE_XS_A + 16)
#define BE_XS_BUF ((BE_XS_LEN > BE_MAX_LM + 16 ? BE_XS_LEN : BE_MAX_LM + 16))      // doubles: the exchange buffers also carry the final gather of the inverse depths (cap <= BE_MAX_LM)
// the summed coefficients as be_solve_shard_kernel reads them (per linearisation set): full 66 x 66 images of S, C2 (rho^2/alpha), C6 (h rho^2), C9 (rho^2) | 7 vectors (V(1..7)) | the scalars
#define BE_QF_M(k) ((k) * 4356)
#define BE_QF_V(k) (4 * 4356 + (k) * 66)
#define BE_QF_A (4 * 4356 + 7 * 66)
#define BE_QF_LEN (BE_QF_A + 16)
struct BeShard { int32_t on, rank, world, lo, hi, cap, len, pad; double* xsend; const double* xrecv; double* qf[2]; };

// Free extrinsic / td blocks in the window solve (estimate_extrinsic 1, estimate_td 1: Estimator::AddBodyParameterBlock, estimator.cpp:87-100 — para_ex_pose[0..1] stay
// variable once openExEstimation is set, para_td while |Vs[0]| >= 0.2).  No shipped YAML switches them on, so they live BESIDE the default path instead of inside it: the
// packets, be_eval, be_reduce and the MF16 solve are untouched (same bits, same speed); with a free block the solve takes the generic factorisation (n up to 178) and two
// more launches per linearisation fill the 13 extra columns of the reduced system:
//   be_eval_ext    per landmark: the factors again WITH their extrinsic / td Jacobians (proj_factor<true, true>) -> one "ext packet" per landmark, transposed like the packets:
//                  ux[13] = J_x^T J_l | gx[13] = J_x^T r | XX[13][13] = J_x^T J_x | XP[13][66] = J_x^T J_pose      (x = ex0 (6), ex1 (6), td (1): local index q)
//   be_reduce_ext  behind be_reduce (which has written the prior's part of every non pose x pose entry): adds the landmark sums into rows / columns xcol[q] of Hd, Sc, gvec
// be_solve (generic form only) carries the 13 extra entries through every product with the landmarks' coupling rows and forms the candidate's ex / td blocks.
#define BE_NX 13
#define BX_W 0
#define BX_G 13
#define BX_XX 26
#define BX_XP 195
#define BX_SIZE 1056              // 1053 used
struct BeExt { int32_t on, pad; int32_t xcol[BE_NX + 1]; double* xpk[2]; };      // xcol[q]: column of ext entry q in the reduced system, -1 = that block is constant

struct BeEvalArgs {
    const BeCtl* ctl;
    const BeState* x; const BeState* cand;
    const BeFactor* fac; const BeLm* lm; const BeImu* imu; const BePriorHdr* prior; const double* priorA; const double* priorb;
    BeDims dims; double g_norm;
    // the linearisation exists twice: set ctl->cur belongs to x, the other one receives the SPECULATIVE linearisation at the candidate
    double* packets[2];     // [nlm][BE_PK_SIZE]                   written by the full evaluation
    double* imu_out[2];     // [nimu][936]: cost, g[30], H[30][30]
    double* prior_out[2];   // cost, g[n_prior]
    double* cand_cost;      // [nlm + nimu + 1]                    written by the evaluations at cand
    int32_t* lm_obs;        // [nlm] bit f: the landmark's factors touch pose f (its anchor or an observing frame).  The packet slots GP / DD / DA of the other
                            // frames are neither written nor read: be_reduce walks, per pose pair, only the landmarks that carry both bits
    int32_t lm_lo, lm_hi;   // landmark blocks outside [lm_lo, lm_hi) return at once (sharded window); 0, nlm otherwise
    const double* prior_c0; // the prior's constant r0^T r0, device resident (the marginalization of the previous frame may still be writing it
                            // when this solve is being prepared on the host)
};

struct BeSolveArgs {
    BeCtl* ctl;
    BeState* x; BeState* cand;
    const BeLm* lm; const BeImu* imu; const BePriorHdr* prior; const double* priorA;
    BeDims dims;
    const double* packets[2]; const double* imu_out[2]; const double* prior_out[2]; const double* cand_cost; const int32_t* lm_obs;
    double* Hd[2];          // [n][n] everything except the Schur term
    double* Sc[2];          // Hd - sum_l rho_l w_l w_l^T (the Schur complement before scaling), block-packed lower triangle (blk_pos in be_solve.hip)
    double* gvec[2];        // [n] gradient g_p ; [n..2n) Schur part sum_l rho_l w_l g_l
    double* scale_p; double* diag_p; double* grad_p; double* gn_p;     // [n]
    double* scale_l; double* diag_l; double* grad_l; double* gn_l;     // [nlm]
    int32_t* prior_col;     // [BE_MAX_STATE] prior index of each state column (-1 if absent)
    int32_t* col_kind; int32_t* col_frame; int32_t* col_comp;          // [n]
    uint8_t ldl_col0[64]; int32_t ldl_mf16, ldl_pad;      // ldl_mf16 != 0: the 16-wide MFMA factorisation, its tile plan (be_mf16_plan, [16 waves][4 slots]) in ldl_col0; 0: the generic 4-wide panel form
    BeShard sh;             // landmark sharding (on = 0: the whole window lives here)
    double xnorm2_extra;    // squared norm of inert free blocks (line blocks under zero sqrt_info) that count in the parameter-tolerance test
    BeExt xt;               // free extrinsic / td blocks (on = 0: constant, the default)
};

struct BeMargArgs {
    const BeState* x; int nframes, nlm, nimu;
    const BeFactor* fac; const BeLm* lm; const BeImu* imu;
    const BePriorHdr* prior; const double* priorA; const double* priorb;
    const int32_t* prior_map;     // [n_old]  old prior index -> dim of the marginalization system, -1 = not present
    const int32_t* imu_map;       // [30]     IMU factor local dim -> dim
    const int32_t* dim_slot;      // [D]      0..10 pose frame, 11 ex0, 12 ex1, 13 td, -1 speed-bias
    const int32_t* dim_comp;      // [D]
    int D, m; double g_norm;
    double* outA; double* outb; double* out_scalars;     // n x n, n, {c0, min pivot, failure flag, rank}
    double* W; double* part; double* psum;                // be_marg_lm -> be_marg_sum: w_l | g_l | 1 / h_l per landmark [nlm][be_marg_wstride(D)]; structured block sums per landmark chunk [chunks][be_marg_part()]; their sum over chunks
    int pose_dim[BE_NF], ex_dim[2], td_dim;               // first dim of each block in the system, -1 = absent (what dim_slot / dim_comp tabulate)
    int c0_mode;                                           // be_marg_finish: 0 everything; 1 all but c0 (A', b', pivot health); 2 c0 only, from A', b' in outA / outb (the side-stream launch)
    double* sum; double* lm_h; int anchor;                // A_lm | b_lm dense (D*D + D); per-landmark h; the frame the landmarks are anchored in (= the dropped one)
    double* imu_w;                // [465] whitened Jacobian (15 x 30) and residual (15) of the IMU factor (0,1): written by the extra block of be_marg_lm, read by be_marg_finish
    double* c0_out;               // optional second home of c0 (the device-resident prior of the estimator)
    const int32_t* lm_sel;        // optional: landmark b of the launch is lm[lm_sel[b]] (marginalization straight out of the solved window)
    // be_marg_finish on the matrix cores (round 5): the dropped block's elimination and the 82-pivot factorisation behind c0 as ONE 16-wide LDL^T (be_mf16.h) of the system
    // [dropped dims, padded to whole tiles | kept dims | right-hand side]: mf16 != 0 and its tile plan (be_mf16_plan for mf_n = 16 ceil(m / 16) + D - m) in mf_plan
    uint8_t mf_plan[64]; int32_t mf16, mf_n;
};

// yaw-gauge fix after a solve (Estimator::Double2vector, estimator.cpp:1111-1154): rotates the solved window back to the yaw and
// position frame 0 had before the solve, in place on the device, so that the marginalization can follow without a host round trip
struct BeGaugeArgs {
    const BeState* x; BeState* out; int nframes, use_imu, nlm; double R0[9]; double ypr0[3]; double P0[3];      // out != x: the raw solution stays intact
    // optional download by the kernel itself (pinned host memory, written over PCIe by the workgroup's last phase): the gauge-fixed state, the control
    // block and the raw poses reach the host with the kernel's completion instead of through three copy dispatches behind it (~15 us of the BA stream
    // and of the host's wake-up per frame)
    BeState* h_out; BeCtl* h_ctl; const BeCtl* ctl; double* h_raw_pose; int state_doubles, pad;
};

// OutliersRejection (vio_util.cpp:381-430) on the device, behind the gauge fix: mean reprojection error of every landmark of the problem over its residual blocks at
// the gauge-fixed states; flag[l] = 1 if it exceeds 3 px (x focal).  The same expressions in the same order as the estimator's host loop (est_host.hip
// reject_outliers): bit-identical decisions.  ric / tic: the camera extrinsics as the host will hold them after Double2vector.
struct BeRejectArgs { const BeState* st; const BeFactor* fac; const BeLm* lm; int nlm, nframes; double ric[2][9], tic[2][3], focal; uint8_t* flags; int ex_from_state, pad; };      // ex_from_state: free extrinsics — ric / tic are the SOLVED blocks of st (what Double2vector hands the host), not the arguments
void be_launch_reject(const BeRejectArgs& a, hipStream_t s);

#if defined(__HIPCC__)
#include "wave_dpp.h"
using namespace be;
__device__ inline void be_frame_geom_dev(const BeState* s, int nframes, FrameGeom* fg, m33* ric, d3* tic, int lane) {
    if (lane < nframes) { fg[lane].R = qR(Q4(s->pose[lane])); fg[lane].P = P3(s->pose[lane]); }
    if (lane >= 32 && lane < 34) { ric[lane - 32] = qR(Q4(s->ex[lane - 32])); tic[lane - 32] = P3(s->ex[lane - 32]); }
}

// dx of the prior's kept blocks (MarginalizationFactor::Evaluate, marginalization_factor.cpp:355-378)
__device__ inline void be_prior_dx_dev(const BePriorHdr* p, const BeState* s, double* dx, int lane, int nthreads) {
    for (int b = lane; b < p->nblocks; b += nthreads) {
        const BePriorBlock pb = p->blocks[b];
        const double* x0 = p->x0[b];
        if (pb.type == 0 || pb.type == 2) {
            const double* x = pb.type == 0 ? s->pose[pb.idx] : s->ex[pb.idx];
            for (int k = 0; k < 3; ++k) dx[pb.off + k] = x[k] - x0[k];
            const quat dq = qmul(qinv(Q4(x0)), Q4(x));
            d3 v = qvec(dq) * 2.0;
            if (!(dq.w >= 0)) v = -v;
            dx[pb.off + 3] = v.x; dx[pb.off + 4] = v.y; dx[pb.off + 5] = v.z;
        } else if (pb.type == 1) {
            for (int k = 0; k < 9; ++k) dx[pb.off + k] = s->sb[pb.idx][k] - x0[k];
        } else dx[pb.off] = s->td - x0[0];
    }
}

// accept / reject decision on the pending candidate (be_accept_kernel; also the first half of the fused accept + gauge kernel of the estimator path)
__device__ __forceinline__ void be_accept_body(const BeSolveArgs& a) {
    BeCtl* ctl = a.ctl;
    const BeCtl c = *ctl;
    __shared__ double red[4];
    __shared__ int s_accept;
    const int tid = threadIdx.x;
    // EVERY wave must hold its copy of the control block before thread 0 stores into it below (pending = 0, slots): without this barrier a wave that starts late — a busy CU
    // staggers the waves of a workgroup by microseconds — loaded the block AFTER that store, saw pending == 0 and left through the return below while wave 0 went on to sum
    // red[] entries nobody had written and to copy a quarter of the state.  Found in round 4 as single members of multi-group runs leaving their trajectory.
    __syncthreads();
    if (c.done || !c.pending) return;          // failed factorisations and invalid steps are settled by the solve kernel itself (workgroup-uniform: every thread holds the same c)
    if (tid == 0) { ctl->slots = c.slots + 1; ctl->pending = 0; s_accept = 0; }
    const int iter = c.iter + 1;
    // candidate cost: fixed-order sum
    const int ncost = a.dims.nlm + a.dims.nimu + 1;
    double part = 0;
    {   // <= 4 costs per thread (BE_MAX_LM + BE_WIN + 1 <= 1024), requested together and added in index order (a load + wait per trip otherwise)
        double cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cv[u] = a.cand_cost[tid + 256 * u < ncost ? tid + 256 * u : 0];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (tid + 256 * u < ncost) part += cv[u];
    }
    part = wave_sum_f64(part);          // the same tree as the decision in be_solve's prologue, bit for bit
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


#endif

// Iteration schedule (be_api.hip): the classic slot is  eval(x) -> reduce -> solve -> eval-cost(cand) -> accept.  The speculative slot
// linearises AT THE CANDIDATE instead (full evaluation + reduce into the other set, with the mu an accepted step would leave) and lets the
// next solve kernel take the accept / reject decision in its prologue: 3 launches per iteration instead of 5, and an accepted step
// (the common case) finds its reduced system ready.  A rejected step costs one wasted reduce.
#define BE_EVAL_X 0         // full evaluation at x into set cur            (runs if need_eval)
#define BE_EVAL_CAND_COST 1 // costs only at cand                           (runs if pending)
#define BE_EVAL_CAND_FULL 2 // full evaluation at cand into set cur ^ 1 + costs (runs if pending)
void be_launch_eval(const BeEvalArgs& a, int mode, hipStream_t s);
int  be_launch_marg(const BeMargArgs& a, hipStream_t s);
int  dv_warm_stream(hipStream_t s);      // copy.hip: one dispatch with ~150 B of scratch per lane (the queue's scratch memory is allocated at create time, not at the first window solve)
int  be_eval_prepare(); int be_solve_prepare(); int be_marg_prepare(); int dv_copy_prepare();      // load the code objects / set the LDS attributes at create time (be_prepare)
int  be_launch_marg_c0(const BeMargArgs& a, hipStream_t s);      // the c0 = b'^T A'^+ b' part alone (BeMargArgs::c0_mode is set to 2)
int  be_marg_chunks(int nlm);      // workgroups of be_marg_lm for nlm landmarks
int  be_marg_part();               // doubles per chunk in BeMargArgs::part
int  be_marg_wstride(int D);       // doubles per landmark in BeMargArgs::W
void be_launch_gauge(const BeGaugeArgs& a, hipStream_t s);
void be_launch_accept_gauge(const BeSolveArgs& sa, const BeGaugeArgs& ga, hipStream_t s);      // be_accept + be_gauge in one launch (estimator path)
void be_launch_reduce(const BeSolveArgs& a, int spec, hipStream_t s);      // spec: reduce the candidate's set (or, after a failed / invalid step, rebuild x's with the new mu)
static_assert(offsetof(BeSolveArgs, ldl_col0) % 4 == 0, "the MF16 plan is read as dwords");
bool be_mf16_plan(int n, uint8_t* plan, bool check_solve_lds = true);      // false: the tiles of an n x n system (+ right-hand-side row) do not fit (n > 175): the generic form is used
int  be_launch_solve(const BeSolveArgs& a, int spec, hipStream_t s);        // spec: decide on the pending candidate first (be_accept_kernel's rule)
void be_launch_accept(const BeSolveArgs& a, hipStream_t s);
void be_launch_eval_ext(const BeEvalArgs& a, const BeExt& xt, int mode, hipStream_t s);      // behind be_launch_eval (full modes only)
void be_launch_reduce_ext(const BeSolveArgs& a, int spec, hipStream_t s);                    // behind be_launch_reduce
// batched forms: n_win independent windows per launch (argument tables in HBM, window index in the grid); be_api.hip enqueues them for a dv_batch
void be_launch_eval_batch(const BeEvalArgs* tab_dev, int n_win, int max_grid, int mode, hipStream_t s);
void be_launch_reduce_batch(const BeSolveArgs* tab_dev, int n_win, int max_n, int spec, hipStream_t s);
int  be_launch_solve_batch(const BeSolveArgs* tab_dev, int n_win, int max_n, int spec, hipStream_t s);      // every window on the MF16 form
void be_launch_accept_batch(const BeSolveArgs* tab_dev, int n_win, hipStream_t s);
int be_eval_batch_blocks(int nlm, int nimu);      // workgroups of one window in the batched evaluation launch
void be_launch_accept_gauge_batch(const BeSolveArgs* stab, const BeGaugeArgs* gtab, int n, hipStream_t s);
void be_launch_reject_batch(const BeRejectArgs* tab, int n, int max_nlm, hipStream_t s);
size_t be_marg_finish_smem(int D, int n);
int be_launch_marg_batch(const BeMargArgs* tab, int n, int max_nlm, int any_imu, int max_D, size_t max_finish_bytes, hipStream_t s);
void be_launch_shard_finalize(const BeSolveArgs& a, int spec, hipStream_t s);   // after the exchange of a reduce: rank-ordered sums -> Hd / Sc / gvec and the form coefficients qf
void be_launch_shard_cost(const BeSolveArgs& a, int phase, hipStream_t s);      // cost-only exchange: phase 0 sums the owned candidate costs into one double, phase 1 leaves the rank-ordered total in cand_cost
void be_launch_shard_depth(const BeSolveArgs& a, int phase, hipStream_t s);     // behind the last slot: phase 0 packs the owned inverse depths of x, phase 1 scatters the gathered ranges
void be_launch_proj_op(const BeFactor* fac, int n, const double* pose_i, const double* pose_j, const double* ex0, const double* ex1,
                       const double* lambda, const double* td, double* out, hipStream_t s);
void be_launch_imu_op(const BeImu* m, double g_norm, const double* par, double* out, hipStream_t s);
