// be_kernels.h — argument blocks and launchers of the back-end kernels (product code, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "be_factor_dev.h"

#define IMU_OUT_STRIDE 936

struct BeEvalArgs {
    const BeCtl* ctl;
    const BeState* x; const BeState* cand;
    const BeFactor* fac; const BeLm* lm; const BeImu* imu; const BePriorHdr* prior; const double* priorA; const double* priorb;
    BeDims dims; double g_norm;
    double* packets;        // [nlm][BE_PK_SIZE]                   written by the full evaluation at x
    double* imu_out;        // [nimu][936]: cost, g[30], H[30][30]
    double* prior_out;      // cost, g[n_prior]
    double* cand_cost;      // [nlm + nimu + 1]                    written by the cost-only evaluation at cand
};

struct BeSolveArgs {
    BeCtl* ctl;
    BeState* x; BeState* cand;
    const BeLm* lm; const BeImu* imu; const BePriorHdr* prior; const double* priorA;
    BeDims dims;
    const double* packets; const double* imu_out; const double* prior_out; const double* cand_cost;
    double* Hd;             // [n][n] everything except the Schur term
    double* Sc;             // [n][n] sum_l rho_l w_l w_l^T
    double* gvec;           // [n] gradient g_p ; [n..2n) Schur part sum_l rho_l w_l g_l
    double* scale_p; double* diag_p; double* grad_p; double* gn_p;     // [n]
    double* scale_l; double* diag_l; double* grad_l; double* gn_l;     // [nlm]
    int32_t* prior_col;     // [BE_MAX_STATE] prior index of each state column (-1 if absent)
    int32_t* col_kind; int32_t* col_frame; int32_t* col_comp;          // [n]
};

void be_launch_eval(const BeEvalArgs& a, bool full, hipStream_t s);
void be_launch_reduce(const BeSolveArgs& a, hipStream_t s);
int  be_launch_solve(const BeSolveArgs& a, hipStream_t s);
void be_launch_accept(const BeSolveArgs& a, hipStream_t s);
void be_launch_proj_op(const BeFactor* fac, int n, const double* pose_i, const double* pose_j, const double* ex0, const double* ex1,
                       const double* lambda, const double* td, double* out, hipStream_t s);
void be_launch_imu_op(const BeImu* m, double g_norm, const double* par, double* out, hipStream_t s);
