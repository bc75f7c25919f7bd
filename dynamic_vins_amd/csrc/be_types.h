// be_types.h — HBM layout of one sliding-window bundle-adjustment problem (product code).
//
// The reference hands Ceres a graph of heap-allocated cost functions (estimator/estimator.cpp:109-214).
// Here the same problem is a handful of flat fp64 tables uploaded once per solve:
//   BeFactor[F]   one 112-byte record per reprojection residual block (the constructor arguments of
//                 ProjectionTwoFrameOneCam/TwoFrameTwoCam/OneFrameTwoCamFactor), grouped by landmark;
//   BeLm[L]       per landmark: first factor, count, anchor frame, observing-frame bitmask;
//   BeImu[<=10]   per IMUFactor: pre-integrated deltas, bias Jacobian blocks, cached sqrt-information;
//   BePrior       the marginalization prior in information form (A', b', c0, linearisation point);
//   BeState       the parameter blocks (para_pose / para_speed_bias / para_ex_pose / para_td /
//                 para_point_features of estimator/body.h:81-87).
// Evaluation writes one fixed-size "packet" per landmark (its Hessian pieces before Schur elimination);
// the reduced camera system is assembled from packets in a fixed order -> bitwise reproducible.
#pragma once
#include <stdint.h>

#define BE_WIN 10                 // kWinSize
#define BE_NF (BE_WIN + 1)        // frames in the window
#define BE_MAX_LM 1000            // kNumFeat (body.h:84, Q26)
#define BE_MAX_OBS_FACTORS 24     // <= 2*11 factors per landmark
#define BE_MAX_STATE 178          // 11*6 + 11*9 + 2*6 + 1
#define BE_MAX_PRIOR 192

struct BeFactor {
    double pix, piy, pjx, pjy;    // pts_i, pts_j (normalised plane, z = 1)
    double vix, viy, vjx, vjy;    // velocity_i, velocity_j
    double td_i, td_j;
    int32_t kind;                 // 0 two-frame one-cam, 1 two-frame two-cam, 2 one-frame two-cam
    int32_t lm, fi, fj;
    double pad_[2];
};
static_assert(sizeof(BeFactor) == 112, "BeFactor is 112 bytes (SURVEY 8(d): algorithmic bytes per residual block)");

struct BeLm { int32_t first, count, anchor, mask; };

struct BeImu {
    double sum_dt;
    double dp[3], dq[4] /* w x y z */, dv[3];
    double lin_ba[3], lin_bg[3];
    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    double sqrt_info[225];        // upper-triangular U with U^T U = covariance^-1 (row-major)
    int32_t fi, fj, pad0, pad1;
};

struct BeState {
    double pose[BE_NF][7];        // x y z qx qy qz qw
    double sb[BE_NF][9];          // V, Ba, Bg
    double ex[2][7];
    double td;
    double inv_depth[BE_MAX_LM];
};

// per-landmark evaluation packet (doubles):
//   [0] h = J_l^T J_l   [1] g_l = J_l^T r   [2] cost
//   [3 + a*6 + r]            w[a][r]   = (J_a^T J_l)          a = frame 0..10
//   [69 + a*6 + r]           gp[a][r]  = (J_a^T r)
//   [135 + a*36 + r*6 + c]   Ddiag[a]  = J_a^T J_a
//   [531 + a*36 + r*6 + c]   Danch[a]  = J_anchor^T J_a  (a != anchor)
#define BE_PK_H 0
#define BE_PK_G 1
#define BE_PK_COST 2
#define BE_PK_W 3
#define BE_PK_GP 69
#define BE_PK_DD 135
#define BE_PK_DA 531
#define BE_PK_SIZE 928            // 927 used, padded to a multiple of 16 bytes
// Packets are stored TRANSPOSED in HBM: entry e of landmark l lives at packets[e * BE_PK_STRIDE + l], so that the
// consumers (one thread per landmark, or one thread per matrix entry looping over landmarks) read coalesced.
#define BE_PK_STRIDE 1024
#define BE_PK(pk, e, l) (pk)[(size_t)(e) * BE_PK_STRIDE + (l)]

struct BePriorBlock { int32_t type /*0 pose,1 sb,2 ex,3 td*/, idx, off, size_local; };

struct BePriorHdr {
    int32_t valid, n, nblocks, pad;
    double c0;                    // r0^T r0
    BePriorBlock blocks[16];
    double x0[16][9];             // linearisation point of every kept block (global size)
};

// solver control block (device resident; every kernel of the iteration schedule reads it first)
struct BeCtl {
    int32_t done, need_eval, reuse, iter, invalid, termination, successful, first, max_iters, chol_fail, step_valid, slots;
    int32_t cur;          // which of the two linearisation sets (packets, IMU / prior blocks, Hd, Sc, gvec) belongs to x
    int32_t pending;      // a valid candidate has been produced and awaits its accept / reject decision
    int32_t alpha_valid;  // the Cauchy point of the current linearisation has been computed (it is evaluated lazily)
    int32_t pad_;
    double radius, mu, x_cost, cand_cost, model_cost_change, dogleg_norm, alpha, x_norm, initial_cost, step_norm;
};

struct BeDims {
    int32_t nframes, nlm, nfac, nimu, nstate, use_imu, plane_kind, pad;
    int32_t pose_col[BE_NF], sb_col[BE_NF];      // column of each block in the reduced system, -1 = constant/absent
};
