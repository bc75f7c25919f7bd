// be_api.hip — host side of the bundle-adjustment entry points of include/dvins.h: uploads the flat problem
// tables, enqueues the fixed kernel schedule of the trust-region loop on the ctx's BA stream (no host round trip
// between iterations: every kernel is predicated on the device-resident BeCtl) and downloads the solved states.
#include "dv_ctx.h"
#include "be_kernels.h"

static_assert(sizeof(dv_ba_factor) == sizeof(BeFactor), "public/private factor layouts must match");
static_assert(sizeof(dv_ba_lm) == sizeof(BeLm), "public/private landmark layouts must match");
static_assert(sizeof(dv_ba_prior) == sizeof(BePriorHdr), "public/private prior layouts must match");

static int be_ensure(dv_ctx* ctx, int nfac) {
    BeWork& w = ctx->be;
    if (w.ready && nfac <= w.fac_cap) return 0;
    const int fac_cap = std::max(nfac, 8192);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t n = BE_MAX_STATE;
    size_t o_ctl = take(sizeof(BeCtl)), o_x = take(sizeof(BeState)), o_c = take(sizeof(BeState)), o_fac = take(sizeof(BeFactor) * (size_t)fac_cap),
           o_lm = take(sizeof(BeLm) * BE_MAX_LM), o_imu = take(sizeof(BeImu) * BE_WIN), o_pr = take(sizeof(BePriorHdr)),
           o_pA = take(8 * (size_t)BE_MAX_PRIOR * BE_MAX_PRIOR), o_pb = take(8 * BE_MAX_PRIOR),
           o_pk = take(8 * (size_t)BE_MAX_LM * BE_PK_SIZE), o_io = take(8 * (size_t)BE_WIN * IMU_OUT_STRIDE), o_po = take(8 * (BE_MAX_PRIOR + 1)),
           o_cc = take(8 * (BE_MAX_LM + BE_WIN + 1)), o_hd = take(8 * n * n), o_sc = take(8 * n * n), o_g = take(8 * 2 * n),
           o_v = take(8 * 4 * n), o_vl = take(8 * 4 * (size_t)BE_MAX_LM), o_i = take(4 * 4 * n);
    DV_CHECK(w.block.ensure(off));
    uint8_t* b = (uint8_t*)w.block.p;
    w.ctl = (BeCtl*)(b + o_ctl); w.x = (BeState*)(b + o_x); w.cand = (BeState*)(b + o_c); w.fac = (BeFactor*)(b + o_fac); w.lm = (BeLm*)(b + o_lm);
    w.imu = (BeImu*)(b + o_imu); w.prior = (BePriorHdr*)(b + o_pr); w.priorA = (double*)(b + o_pA); w.priorb = (double*)(b + o_pb);
    w.packets = (double*)(b + o_pk); w.imu_out = (double*)(b + o_io); w.prior_out = (double*)(b + o_po); w.cand_cost = (double*)(b + o_cc);
    w.Hd = (double*)(b + o_hd); w.Sc = (double*)(b + o_sc); w.gvec = (double*)(b + o_g);
    double* v = (double*)(b + o_v); w.scale_p = v; w.diag_p = v + n; w.grad_p = v + 2 * n; w.gn_p = v + 3 * n;
    double* vl = (double*)(b + o_vl); w.scale_l = vl; w.diag_l = vl + BE_MAX_LM; w.grad_l = vl + 2 * BE_MAX_LM; w.gn_l = vl + 3 * BE_MAX_LM;
    int32_t* iv = (int32_t*)(b + o_i); w.prior_col = iv; w.col_kind = iv + n; w.col_frame = iv + 2 * n; w.col_comp = iv + 3 * n;
    w.fac_cap = fac_cap;
    if (!w.pinned) {
        w.pinned_bytes = sizeof(BeState) + sizeof(BeCtl) + sizeof(BeImu) * BE_WIN + 4096;
        DV_CHECK(hipHostMalloc(&w.pinned, w.pinned_bytes, hipHostMallocDefault));
    }
    w.ready = true;
    return 0;
}

// 15x15: U upper-triangular with U^T U = cov^-1   (LLT(cov^-1).matrixL().transpose(), imu_factor.h:74-75; cached, Q8)
static bool imu_sqrt_info(const double* cov, double* U) {
    double a[15][30];
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { a[i][j] = cov[i * 15 + j]; a[i][15 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 15; ++c) {
        int p = c; for (int i = c + 1; i < 15; ++i) if (std::fabs(a[i][c]) > std::fabs(a[p][c])) p = i;
        if (a[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 30; ++j) std::swap(a[c][j], a[p][j]);
        const double piv = a[c][c];
        for (int j = 0; j < 30; ++j) a[c][j] /= piv;
        for (int i = 0; i < 15; ++i) if (i != c) { const double f = a[i][c]; if (f != 0.0) for (int j = 0; j < 30; ++j) a[i][j] -= f * a[c][j]; }
    }
    double inv[15][15], L[15][15] = { { 0 } };
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) inv[i][j] = 0.5 * (a[i][15 + j] + a[j][15 + i]);
    for (int j = 0; j < 15; ++j) {
        double s = inv[j][j];
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        if (!(s > 0)) return false;
        const double d = std::sqrt(s); L[j][j] = d;
        for (int i = j + 1; i < 15; ++i) { double t = inv[i][j]; for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k]; L[i][j] = t / d; }
    }
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) U[i * 15 + j] = L[j][i];
    return true;
}

int be_fill_imu(const dv_ba_imu& in, BeImu& o) {
    o.sum_dt = in.sum_dt;
    for (int k = 0; k < 3; ++k) { o.dp[k] = in.dp[k]; o.dv[k] = in.dv[k]; o.lin_ba[k] = in.lin_ba[k]; o.lin_bg[k] = in.lin_bg[k]; }
    for (int k = 0; k < 4; ++k) o.dq[k] = in.dq[k];
    auto blk = [&](int r0, int c0, double* dst) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dst[i * 3 + j] = in.jacobian[(r0 + i) * 15 + c0 + j]; };
    blk(0, 9, o.dp_dba); blk(0, 12, o.dp_dbg); blk(3, 12, o.dq_dbg); blk(6, 9, o.dv_dba); blk(6, 12, o.dv_dbg);
    o.fi = in.fi; o.fj = in.fj; o.pad0 = o.pad1 = 0;
    return imu_sqrt_info(in.covariance, o.sqrt_info) ? 0 : -1;
}

extern "C" {

int dv_ba_solve(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary) {
    if (!ctx) return -1;
    if (!P || !P->pose || !P->ex_pose || !P->td) DV_FAIL("dv_ba_solve: null argument");
    if (P->nframes < 1 || P->nframes > BE_NF) DV_FAIL("dv_ba_solve: nframes out of range");
    if (P->nlm < 0 || P->nlm > BE_MAX_LM) DV_FAIL("dv_ba_solve: more than kNumFeat=1000 landmarks");
    if (P->nimu < 0 || P->nimu > BE_WIN) DV_FAIL("dv_ba_solve: bad IMU factor count");
    if (P->use_imu && !P->speed_bias) DV_FAIL("dv_ba_solve: speed_bias is null");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (be_ensure(ctx, P->nfac)) return -1;
    BeWork& w = ctx->be;
    hipStream_t s = ctx->be_stream;
    // ---- column layout of the reduced system ----
    BeDims d{};
    d.nframes = P->nframes; d.nlm = P->nlm; d.nfac = P->nfac; d.nimu = P->nimu; d.use_imu = P->use_imu; d.plane_kind = P->plane_kind;
    std::vector<int32_t> idx(4 * BE_MAX_STATE, -1);
    int32_t* prior_col = idx.data(); int32_t* col_kind = prior_col + BE_MAX_STATE; int32_t* col_frame = col_kind + BE_MAX_STATE; int32_t* col_comp = col_frame + BE_MAX_STATE;
    int col = 0;
    for (int f = 0; f < BE_NF; ++f) { d.pose_col[f] = -1; d.sb_col[f] = -1; }
    for (int f = 0; f < P->nframes; ++f) {
        const bool pose_const = !P->use_imu && f == 0;              // estimator.cpp:83-84
        if (!pose_const) { d.pose_col[f] = col; for (int k = 0; k < 6; ++k) { col_kind[col] = 0; col_frame[col] = f; col_comp[col] = k; ++col; } }
        if (P->use_imu) { d.sb_col[f] = col; for (int k = 0; k < 9; ++k) { col_kind[col] = 1; col_frame[col] = f; col_comp[col] = k; ++col; } }
    }
    d.nstate = col;
    if (col == 0) DV_FAIL("dv_ba_solve: no free parameter block");
    BePriorHdr ph{};
    const bool has_prior = P->prior && P->prior->valid;
    if (has_prior) {
        std::memcpy(&ph, P->prior, sizeof(ph));
        if (ph.n > BE_MAX_PRIOR || ph.nblocks > 16) DV_FAIL("dv_ba_solve: prior too large");
        for (int b = 0; b < ph.nblocks; ++b) {
            const BePriorBlock pb = ph.blocks[b];
            int c0 = -1;
            if (pb.type == 0) c0 = d.pose_col[pb.idx]; else if (pb.type == 1) c0 = d.sb_col[pb.idx];
            if (c0 >= 0) for (int k = 0; k < pb.size_local; ++k) prior_col[c0 + k] = pb.off + k;
        }
    }
    // ---- upload ----
    BeState* hx = (BeState*)w.pinned;
    std::memset(hx, 0, offsetof(BeState, inv_depth));
    for (int f = 0; f < P->nframes; ++f) { std::memcpy(hx->pose[f], P->pose + 7 * f, 56); if (P->use_imu) std::memcpy(hx->sb[f], P->speed_bias + 9 * f, 72); }
    std::memcpy(hx->ex, P->ex_pose, 14 * 8); hx->td = P->td[0];
    if (P->nlm) std::memcpy(hx->inv_depth, P->inv_depth, 8 * (size_t)P->nlm);
    const size_t state_bytes = offsetof(BeState, inv_depth) + 8 * (size_t)P->nlm;
    BeCtl* hctl = (BeCtl*)((uint8_t*)w.pinned + sizeof(BeState));
    std::memset(hctl, 0, sizeof(BeCtl));
    hctl->need_eval = 1; hctl->first = 1; hctl->max_iters = P->max_iters; hctl->radius = 1e4; hctl->mu = 1e-8; hctl->step_valid = 0;
    BeImu* himu = (BeImu*)((uint8_t*)w.pinned + sizeof(BeState) + sizeof(BeCtl));
    for (int k = 0; k < P->nimu; ++k) if (be_fill_imu(P->imu[k], himu[k])) DV_FAIL("dv_ba_solve: IMU covariance is singular");
    DV_CHECK(hipMemcpyAsync(w.x, hx, state_bytes, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(w.ctl, hctl, sizeof(BeCtl), hipMemcpyHostToDevice, s));
    if (P->nimu) DV_CHECK(hipMemcpyAsync(w.imu, himu, sizeof(BeImu) * P->nimu, hipMemcpyHostToDevice, s));
    if (P->nfac) DV_CHECK(hipMemcpyAsync(w.fac, P->factors, sizeof(BeFactor) * (size_t)P->nfac, hipMemcpyHostToDevice, s));
    if (P->nlm) DV_CHECK(hipMemcpyAsync(w.lm, P->landmarks, sizeof(BeLm) * (size_t)P->nlm, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(w.prior, &ph, sizeof(ph), hipMemcpyHostToDevice, s));
    if (has_prior) {
        DV_CHECK(hipMemcpyAsync(w.priorA, P->prior_A, 8 * (size_t)ph.n * ph.n, hipMemcpyHostToDevice, s));
        DV_CHECK(hipMemcpyAsync(w.priorb, P->prior_b, 8 * (size_t)ph.n, hipMemcpyHostToDevice, s));
    }
    DV_CHECK(hipMemcpyAsync(w.prior_col, idx.data(), 4 * idx.size(), hipMemcpyHostToDevice, s));
    DV_CHECK(hipStreamSynchronize(s));        // idx / ph are host stack/heap objects
    // ---- schedule ----
    BeEvalArgs ea{};
    ea.ctl = w.ctl; ea.x = w.x; ea.cand = w.cand; ea.fac = w.fac; ea.lm = w.lm; ea.imu = w.imu; ea.prior = w.prior; ea.priorA = w.priorA; ea.priorb = w.priorb;
    ea.dims = d; ea.g_norm = P->g_norm; ea.packets = w.packets; ea.imu_out = w.imu_out; ea.prior_out = w.prior_out; ea.cand_cost = w.cand_cost;
    BeSolveArgs sa{};
    sa.ctl = w.ctl; sa.x = w.x; sa.cand = w.cand; sa.lm = w.lm; sa.imu = w.imu; sa.prior = w.prior; sa.priorA = w.priorA; sa.dims = d;
    sa.packets = w.packets; sa.imu_out = w.imu_out; sa.prior_out = w.prior_out; sa.cand_cost = w.cand_cost; sa.Hd = w.Hd; sa.Sc = w.Sc; sa.gvec = w.gvec;
    sa.scale_p = w.scale_p; sa.diag_p = w.diag_p; sa.grad_p = w.grad_p; sa.gn_p = w.gn_p; sa.scale_l = w.scale_l; sa.diag_l = w.diag_l; sa.grad_l = w.grad_l; sa.gn_l = w.gn_l;
    sa.prior_col = w.prior_col; sa.col_kind = w.col_kind; sa.col_frame = w.col_frame; sa.col_comp = w.col_comp;
    {
        StageScope sc(ctx, "ba_solve", s);
        const int slots = P->max_iters + 3;      // + retries after a failed Cholesky (mu *= 10)
        for (int it = 0; it < slots; ++it) {
            be_launch_eval(ea, true, s);
            be_launch_reduce(sa, s);
            if (be_launch_solve(sa, s)) DV_FAIL("dv_ba_solve: cannot set dynamic LDS size");
            be_launch_eval(ea, false, s);
            be_launch_accept(sa, s);
        }
    }
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(hx, w.x, state_bytes, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(hctl, w.ctl, sizeof(BeCtl), hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (ctx->timing) dv_harvest_timers(ctx);
    for (int f = 0; f < P->nframes; ++f) { std::memcpy(P->pose + 7 * f, hx->pose[f], 56); if (P->use_imu) std::memcpy(P->speed_bias + 9 * f, hx->sb[f], 72); }
    if (P->nlm) std::memcpy(P->inv_depth, hx->inv_depth, 8 * (size_t)P->nlm);
    if (summary) {
        summary->iterations = hctl->iter; summary->successful = hctl->successful; summary->termination = hctl->done ? hctl->termination : 0;
        summary->slots = hctl->slots; summary->initial_cost = hctl->initial_cost; summary->final_cost = hctl->x_cost;
    }
    return 0;
}

int dv_proj_eval(dv_ctx* ctx, const dv_ba_factor* factors, int n, const double* pose_i, const double* pose_j, const double* ex0,
                 const double* ex1, const double* inv_depth, const double* td, double* out) {
    if (!ctx) return -1;
    if (!factors || n <= 0 || !out) DV_FAIL("dv_proj_eval: bad argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->be_stream;
    const size_t nf = sizeof(BeFactor) * (size_t)n, np = 56 * (size_t)n, ns = 8 * (size_t)n, no = 8 * 54 * (size_t)n;
    DV_CHECK(ctx->s0.ensure(nf + 4 * np + 2 * ns + no + 1024));
    uint8_t* b = (uint8_t*)ctx->s0.p;
    BeFactor* dfac = (BeFactor*)b; double* dpi = (double*)(b + nf); double* dpj = dpi + 7 * n; double* de0 = dpj + 7 * n; double* de1 = de0 + 7 * n;
    double* dl = de1 + 7 * n; double* dtd = dl + n; double* dout = dtd + n;
    DV_CHECK(hipMemcpyAsync(dfac, factors, nf, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(dpi, pose_i, np, hipMemcpyHostToDevice, s)); DV_CHECK(hipMemcpyAsync(dpj, pose_j, np, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(de0, ex0, np, hipMemcpyHostToDevice, s)); DV_CHECK(hipMemcpyAsync(de1, ex1, np, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(dl, inv_depth, ns, hipMemcpyHostToDevice, s)); DV_CHECK(hipMemcpyAsync(dtd, td, ns, hipMemcpyHostToDevice, s));
    be_launch_proj_op(dfac, n, dpi, dpj, de0, de1, dl, dtd, dout, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, no, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    return 0;
}

int dv_imu_eval(dv_ctx* ctx, const dv_ba_imu* imu, double g_norm, const double* pose_i, const double* sb_i, const double* pose_j,
                const double* sb_j, double* out) {
    if (!ctx) return -1;
    if (!imu || !out) DV_FAIL("dv_imu_eval: bad argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->be_stream;
    BeImu h;
    if (be_fill_imu(*imu, h)) DV_FAIL("dv_imu_eval: IMU covariance is singular");
    DV_CHECK(ctx->s0.ensure(sizeof(BeImu) + 8 * (32 + 465) + 256));
    uint8_t* b = (uint8_t*)ctx->s0.p;
    BeImu* dm = (BeImu*)b; double* dpar = (double*)(b + sizeof(BeImu)); double* dout = dpar + 32;
    double par[32];
    std::memcpy(par, pose_i, 56); std::memcpy(par + 7, sb_i, 72); std::memcpy(par + 16, pose_j, 56); std::memcpy(par + 23, sb_j, 72);
    DV_CHECK(hipMemcpyAsync(dm, &h, sizeof(h), hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(dpar, par, sizeof(par), hipMemcpyHostToDevice, s));
    be_launch_imu_op(dm, g_norm, dpar, dout, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpyAsync(out, dout, 8 * 465, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
