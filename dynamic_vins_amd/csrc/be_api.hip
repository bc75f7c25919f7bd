// be_api.hip — host side of the bundle-adjustment entry points of include/dvins.h: uploads the flat problem
// tables, enqueues the fixed kernel schedule of the trust-region loop on the ctx's BA stream (no host round trip
// between iterations: every kernel is predicated on the device-resident BeCtl) and downloads the solved states.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include "dv_ctx.h"
#include "be_kernels.h"

static_assert(sizeof(dv_ba_factor) == sizeof(BeFactor), "public/private factor layouts must match");
static_assert(sizeof(dv_ba_lm) == sizeof(BeLm), "public/private landmark layouts must match");
static_assert(sizeof(dv_ba_prior) == sizeof(BePriorHdr), "public/private prior layouts must match");

static int be_ensure(dv_ctx* ctx, int nfac) {
    BeWork& w = ctx->be;
    if (w.ready && nfac <= w.fac_cap) return 0;
    const int fac_cap = std::max(nfac, BE_MAX_LM * BE_MAX_OBS_FACTORS);      // the worst case up front: a re-allocation would drop the device-resident prior
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t n = BE_MAX_STATE;
    // upload region first (mirrored 1:1 by the pinned staging buffer -> ONE H2D copy per solve), factors last (variable length)
    size_t o_ctl = take(sizeof(BeCtl)), o_x = take(sizeof(BeState)), o_imu = take(sizeof(BeImu) * BE_WIN), o_pr = take(sizeof(BePriorHdr)), o_i = take(4 * 4 * n),
           o_mt = take(4 * BE_MARG_TAB_INTS), o_lm = take(sizeof(BeLm) * BE_MAX_LM), o_fac = take(sizeof(BeFactor) * (size_t)fac_cap);
    const size_t upload_bytes = off;
    size_t o_c = take(sizeof(BeState)),
           o_pA = take(8 * (size_t)BE_MAX_PRIOR * BE_MAX_PRIOR), o_pb = take(8 * BE_MAX_PRIOR),
           o_pA2 = take(8 * (size_t)BE_MAX_PRIOR * BE_MAX_PRIOR), o_pb2 = take(8 * BE_MAX_PRIOR), o_ms = take(64), o_c0 = take(64),
           o_pk = take(8 * (size_t)BE_PK_SIZE * BE_PK_STRIDE), o_io = take(8 * (size_t)BE_WIN * IMU_OUT_STRIDE), o_po = take(8 * (BE_MAX_PRIOR + 1)),
           o_cc = take(8 * (BE_MAX_LM + BE_WIN + 1)), o_ob = take(4 * BE_MAX_LM), o_hd = take(8 * n * n), o_sc = take(8 * n * n), o_g = take(8 * 2 * n),
           o_pk1 = take(8 * (size_t)BE_PK_SIZE * BE_PK_STRIDE), o_io1 = take(8 * (size_t)BE_WIN * IMU_OUT_STRIDE), o_po1 = take(8 * (BE_MAX_PRIOR + 1)),
           o_hd1 = take(8 * n * n), o_sc1 = take(8 * n * n), o_g1 = take(8 * 2 * n),      // second linearisation set (speculative evaluation at the candidate)
           o_v = take(8 * 4 * n), o_vl = take(8 * 4 * (size_t)BE_MAX_LM);
    DV_CHECK(w.block.ensure(off));
    uint8_t* b = (uint8_t*)w.block.p;
    w.ctl = (BeCtl*)(b + o_ctl); w.x = (BeState*)(b + o_x); w.cand = (BeState*)(b + o_c); w.fac = (BeFactor*)(b + o_fac); w.lm = (BeLm*)(b + o_lm);
    w.imu = (BeImu*)(b + o_imu); w.prior = (BePriorHdr*)(b + o_pr);
    w.priorA_buf[0] = (double*)(b + o_pA); w.priorb_buf[0] = (double*)(b + o_pb); w.priorA_buf[1] = (double*)(b + o_pA2); w.priorb_buf[1] = (double*)(b + o_pb2);
    w.prior_cur = 0; w.priorA = w.priorA_buf[0]; w.priorb = w.priorb_buf[0]; w.prior_resident = false;
    w.marg_tab = (int32_t*)(b + o_mt); w.marg_scal = (double*)(b + o_ms); w.prior_c0 = (double*)(b + o_c0);
    w.packets[0] = (double*)(b + o_pk); w.imu_out[0] = (double*)(b + o_io); w.prior_out[0] = (double*)(b + o_po); w.cand_cost = (double*)(b + o_cc); w.lm_obs = (int32_t*)(b + o_ob);
    w.Hd[0] = (double*)(b + o_hd); w.Sc[0] = (double*)(b + o_sc); w.gvec[0] = (double*)(b + o_g);
    w.packets[1] = (double*)(b + o_pk1); w.imu_out[1] = (double*)(b + o_io1); w.prior_out[1] = (double*)(b + o_po1);
    w.Hd[1] = (double*)(b + o_hd1); w.Sc[1] = (double*)(b + o_sc1); w.gvec[1] = (double*)(b + o_g1);
    double* v = (double*)(b + o_v); w.scale_p = v; w.diag_p = v + n; w.grad_p = v + 2 * n; w.gn_p = v + 3 * n;
    double* vl = (double*)(b + o_vl); w.scale_l = vl; w.diag_l = vl + BE_MAX_LM; w.grad_l = vl + 2 * BE_MAX_LM; w.gn_l = vl + 3 * BE_MAX_LM;
    int32_t* iv = (int32_t*)(b + o_i); w.prior_col = iv; w.col_kind = iv + n; w.col_frame = iv + 2 * n; w.col_comp = iv + 3 * n;
    w.fac_cap = fac_cap;
    w.up_ctl = o_ctl; w.up_x = o_x; w.up_imu = o_imu; w.up_prior = o_pr; w.up_idx = o_i; w.up_mt = o_mt; w.up_lm = o_lm; w.up_fac = o_fac;
    const size_t need = upload_bytes + sizeof(BeState) + sizeof(BeCtl) + 4096;        // staging mirror + download area
    if (w.pinned_bytes < need) {
        if (w.pinned) (void)hipHostFree(w.pinned);
        w.pinned = nullptr;
        DV_CHECK(hipHostMalloc(&w.pinned, need, hipHostMallocDefault));
        w.pinned_bytes = need;
    }
    w.dl_off = upload_bytes;
    if (!w.ev_state) DV_CHECK(hipEventCreateWithFlags(&w.ev_state, hipEventDisableTiming));
    if (const char* e = std::getenv("DVINS_GPU_REJECT")) w.gpu_reject = std::atoi(e) != 0;
    w.ready = true;
    return 0;
}

// Everything the first frames of an estimator would otherwise create in the middle of the sequence: the work block and its pinned mirror (hipMalloc / hipHostMalloc),
// the marginalization scratch, the side stream of the prior's constant, events, the pinned flag arrays, the kernels' code objects and LDS attributes — together a
// 3 ms frame at the first window solve (scripts/dyn_cold_frames.py), where a 20 Hz estimator has 1 ms frames otherwise.  Called by dv_est_create.
int be_prepare(dv_ctx* ctx, bool dynamic) {
    BeWork& w = ctx->be;
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (be_ensure(ctx, 0)) return -1;
    {       // marg_args' sizing at its maximum (BE_MAX_LM landmarks anchored in the oldest frame, D = 178)
        const size_t nlmax = (size_t)BE_MAX_LM, Dmax = 178;
        DV_CHECK(w.marg_buf.ensure(8 * (nlmax * (size_t)be_marg_wstride((int)Dmax) + ((size_t)be_marg_chunks((int)nlmax) + 1) * be_marg_part() + Dmax * Dmax + Dmax + nlmax + 512)));
    }
    if (w.c0_side && !w.c0_stream) {
        DV_CHECK(hipStreamCreateWithFlags(&w.c0_stream, hipStreamNonBlocking));
        DV_CHECK(hipEventCreateWithFlags(&w.ev_margA, hipEventDisableTiming)); DV_CHECK(hipEventCreateWithFlags(&w.ev_c0, hipEventDisableTiming));
    }
    if (!w.rej_pinned) DV_CHECK(hipHostMalloc((void**)&w.rej_pinned, BE_MAX_LM, hipHostMallocDefault));
    if (be_eval_prepare() || be_solve_prepare() || be_marg_prepare() || dv_copy_prepare()) DV_FAIL("be_prepare: cannot load the back end's kernels");
    if (dynamic && be_obj_solve_prepare(ctx, ctx->obj_buf, ctx->obj_pend)) return -1;
    // the scratch memory of the queues the back end launches on (see dv_warm_stream), then everything above has happened before the first frame
    if (dv_warm_stream(ctx->be_stream) || (w.c0_stream && dv_warm_stream(w.c0_stream)) || (dynamic && ctx->obj_stream && dv_warm_stream(ctx->obj_stream))) DV_FAIL("be_prepare: warm-up launch failed");
    DV_CHECK(hipStreamSynchronize(ctx->be_stream));
    if (w.c0_stream) DV_CHECK(hipStreamSynchronize(w.c0_stream));
    if (dynamic && ctx->obj_stream) DV_CHECK(hipStreamSynchronize(ctx->obj_stream));
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

int be_fill_imu(const dv_ba_imu& in, BeImu& o, const double* sqrt_hint) {
    o.sum_dt = in.sum_dt;
    for (int k = 0; k < 3; ++k) { o.dp[k] = in.dp[k]; o.dv[k] = in.dv[k]; o.lin_ba[k] = in.lin_ba[k]; o.lin_bg[k] = in.lin_bg[k]; }
    for (int k = 0; k < 4; ++k) o.dq[k] = in.dq[k];
    auto blk = [&](int r0, int c0, double* dst) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dst[i * 3 + j] = in.jacobian[(r0 + i) * 15 + c0 + j]; };
    blk(0, 9, o.dp_dba); blk(0, 12, o.dp_dbg); blk(3, 12, o.dq_dbg); blk(6, 9, o.dv_dba); blk(6, 12, o.dv_dbg);
    o.fi = in.fi; o.fj = in.fj; o.pad0 = o.pad1 = 0;
    if (sqrt_hint) { std::memcpy(o.sqrt_info, sqrt_hint, sizeof(o.sqrt_info)); return 0; }
    return imu_sqrt_info(in.covariance, o.sqrt_info) ? 0 : -1;
}
bool be_imu_sqrt_info(const double* cov, double* U) { return imu_sqrt_info(cov, U); }

// ================================ marginalization: structure ("plan"), launch, new header ================================
// Which parameter blocks take part (MarginalizationInfo::addResidualBlockInfo) and where they sit in the dense system:
// dropped dims first, then the kept ones in canonical order (poses, speed-bias, ex0, ex1, td)  (M1).
// sel[0..nsel): landmarks of `lms` whose residual blocks take part (all of them anchored in frame 0).
static int marg_plan(dv_ctx* ctx, MargPlan& pl, int mode, const dv_ba_prior* prior, const dv_ba_factor* fac, const dv_ba_lm* lms, const int* sel, int nsel, bool imu01) {
    pl = MargPlan();
    pl.mode = mode;
    for (int i = 0; i < BE_MARG_TAB_INTS; ++i) pl.tab[i] = -1;
    bool pose_in[BE_NF] = { false }, sb_in[BE_NF] = { false }, ex_in[2] = { false, false }, td_in = false;
    const bool has_prior = prior && prior->valid;
    if (has_prior) for (int b = 0; b < prior->nblocks; ++b) {
        const dv_ba_prior_block& pb = prior->blocks[b];
        if (pb.type == 0) pose_in[pb.idx] = true; else if (pb.type == 1) sb_in[pb.idx] = true; else if (pb.type == 2) ex_in[pb.idx] = true; else td_in = true;
    }
    pl.nimu = (mode == 0 && imu01) ? 1 : 0;
    pl.nsel = (mode == 0) ? nsel : 0;
    if (pl.nimu) { pose_in[0] = sb_in[0] = pose_in[1] = sb_in[1] = true; }
    for (int q = 0; q < pl.nsel; ++q) {
        const dv_ba_lm& L = lms[sel[q]];
        if (L.anchor != 0) DV_FAIL("dv_marginalize: only landmarks anchored in frame 0 take part (estimator.cpp:446)");
        pl.tab[BE_MT_SEL + q] = sel[q];
        for (int f = L.first; f < L.first + L.count; ++f) {
            const dv_ba_factor& ff = fac[f];
            ex_in[0] = true; td_in = true;
            if (ff.kind != 0) ex_in[1] = true;
            if (ff.kind != 2) { pose_in[0] = true; pose_in[ff.fj] = true; }
        }
    }
    int32_t* dim_slot = pl.tab + BE_MT_SLOT; int32_t* dim_comp = pl.tab + BE_MT_COMP;
    int nd = 0;
    for (int k = 0; k < BE_NF; ++k) { pl.pose_dim[k] = -1; pl.sb_dim[k] = -1; }
    pl.ex_dim[0] = pl.ex_dim[1] = -1; pl.td_dim = -1;
    auto add_pose = [&](int k) { pl.pose_dim[k] = nd; for (int c = 0; c < 6; ++c) { dim_slot[nd] = k; dim_comp[nd] = c; ++nd; } };
    auto add_sb = [&](int k) { pl.sb_dim[k] = nd; for (int c = 0; c < 9; ++c) { dim_slot[nd] = -1; dim_comp[nd] = c; ++nd; } };
    const int drop_frame = (mode == 0) ? 0 : BE_WIN - 1;
    if (pose_in[drop_frame]) add_pose(drop_frame);
    if (mode == 0 && sb_in[0]) add_sb(0);
    pl.m = nd;
    if (pl.m == 0) { pl.empty = true; return 0; }             // "unstable tracking" (marginalization_factor.cpp:210-215)
    for (int k = 0; k < BE_NF; ++k) if (pose_in[k] && k != drop_frame) add_pose(k);
    for (int k = 0; k < BE_NF; ++k) if (sb_in[k] && !(mode == 0 && k == 0)) add_sb(k);
    for (int c = 0; c < 2; ++c) if (ex_in[c]) { pl.ex_dim[c] = nd; for (int q = 0; q < 6; ++q) { dim_slot[nd] = BE_NF + c; dim_comp[nd] = q; ++nd; } }
    if (td_in) { pl.td_dim = nd; dim_slot[nd] = BE_NF + 2; dim_comp[nd] = 0; ++nd; }
    pl.D = nd; pl.n = nd - pl.m;
    if (pl.n > BE_MAX_PRIOR || pl.n < 1 || nd > 256) DV_FAIL("dv_marginalize: bad kept size");
    if (has_prior) for (int b = 0; b < prior->nblocks; ++b) {
        const dv_ba_prior_block& pb = prior->blocks[b];
        const int d0 = pb.type == 0 ? pl.pose_dim[pb.idx] : pb.type == 1 ? pl.sb_dim[pb.idx] : pb.type == 2 ? pl.ex_dim[pb.idx] : pl.td_dim;
        for (int k = 0; k < pb.size_local; ++k) pl.tab[BE_MT_PRIOR + pb.off + k] = d0 + k;
    }
    if (pl.nimu) {
        for (int k = 0; k < 6; ++k) { pl.tab[BE_MT_IMU + k] = pl.pose_dim[0] + k; pl.tab[BE_MT_IMU + 15 + k] = pl.pose_dim[1] + k; }
        for (int k = 0; k < 9; ++k) { pl.tab[BE_MT_IMU + 6 + k] = pl.sb_dim[0] + k; pl.tab[BE_MT_IMU + 21 + k] = pl.sb_dim[1] + k; }
    }
    return 0;
}

// the argument block of the three marginalization kernels; the index tables must already be (enqueued to be) in w.marg_tab
static int marg_args(dv_ctx* ctx, const MargPlan& pl, const BeState* x, double g_norm, const double* priorA, const double* priorb, double* outA, double* outb, double* scal, double* c0_out, BeMargArgs& ma) {
    BeWork& w = ctx->be;
    ma = BeMargArgs{};
    ma.x = x; ma.nframes = BE_NF; ma.nlm = pl.nsel; ma.nimu = pl.nimu; ma.fac = w.fac; ma.lm = w.lm; ma.imu = w.imu;
    ma.prior = w.prior; ma.priorA = priorA; ma.priorb = priorb;
    ma.prior_map = w.marg_tab + BE_MT_PRIOR; ma.imu_map = w.marg_tab + BE_MT_IMU; ma.dim_slot = w.marg_tab + BE_MT_SLOT; ma.dim_comp = w.marg_tab + BE_MT_COMP;
    ma.lm_sel = w.marg_tab + BE_MT_SEL;
    ma.D = pl.D; ma.m = pl.m; ma.g_norm = g_norm; ma.outA = outA; ma.outb = outb; ma.out_scalars = scal; ma.c0_out = c0_out;
    const size_t slab = (size_t)pl.D * pl.D + pl.D;
    // sized once for BE_MAX_LM landmarks anchored in the oldest frame at the largest system (D = 178): growing it later would stall the stream
    const size_t nl = (size_t)std::max(pl.nsel, 1), nlmax = std::max(nl, (size_t)BE_MAX_LM), Dmax = (size_t)std::max(pl.D, 178);
    const size_t need = 8 * (nlmax * (size_t)be_marg_wstride((int)Dmax) + ((size_t)be_marg_chunks((int)nlmax) + 1) * be_marg_part() + Dmax * Dmax + Dmax + nlmax + 512);      // W | part | psum | sum | h | whitened IMU factor
    DV_CHECK(w.marg_buf.ensure(need));
    ma.W = (double*)w.marg_buf.p; ma.part = ma.W + nl * be_marg_wstride(pl.D); ma.psum = ma.part + (size_t)be_marg_chunks((int)nl) * be_marg_part(); ma.sum = ma.psum + be_marg_part();
    ma.lm_h = ma.sum + slab; ma.imu_w = ma.lm_h + nl; ma.anchor = 0;
    for (int k = 0; k < BE_NF; ++k) ma.pose_dim[k] = pl.pose_dim[k];
    ma.ex_dim[0] = pl.ex_dim[0]; ma.ex_dim[1] = pl.ex_dim[1]; ma.td_dim = pl.td_dim;
    ma.c0_mode = 0;
    {   // the finish kernel's factorisation on the matrix cores where the tiles fit (every window the estimator builds: D = 97, m = 15 -> 7 x 7 tiles)
        const int mt = (pl.m + 15) / 16, mf_n = 16 * mt + (pl.D - pl.m), NB = (mf_n + 16) >> 4;
        const size_t room = (size_t)pl.D * pl.D + pl.D + std::max((size_t)(pl.D - pl.m) * (pl.D - pl.m), (size_t)1024);      // A | b | W2 of the LDS image: the factor's fragments and the staged A', b' tiles take their place once the tiles are in registers
        ma.mf16 = 0; ma.mf_n = mf_n;
        if (pl.m > 0 && pl.D > pl.m && 2 * ((size_t)NB * (NB + 1) / 2 * 256) <= room && be_mf16_plan(mf_n, ma.mf_plan, false) && !std::getenv("DVINS_MARG_GENERIC")) ma.mf16 = 1;
    }
    return 0;
}
// launches the three kernels
static int marg_enqueue(dv_ctx* ctx, const MargPlan& pl, const BeState* x, double g_norm, const double* priorA, const double* priorb, double* outA, double* outb, double* scal, double* c0_out, hipStream_t s, hipStream_t c0_side = nullptr) {
    BeWork& w = ctx->be;
    BeMargArgs ma;
    if (marg_args(ctx, pl, x, g_norm, priorA, priorb, outA, outb, scal, c0_out, ma)) return -1;
    ma.c0_mode = c0_side ? 1 : 0;
    {
        StageScope sc(ctx, "k_be_marg", s);
        const int rc = be_launch_marg(ma, s);
        if (rc == -2) DV_FAIL("dv_marginalize: system does not fit in LDS");
        if (rc) DV_FAIL("dv_marginalize: cannot set dynamic LDS size");
    }
    if (c0_side) {      // c0 = b'^T A'^+ b' (a third of the marginalization's time) is needed by the next frame's first evaluation only: on a side stream, beside the
                        // host turnaround and the next upload (be_begin_impl waits for ev_c0)
        DV_CHECK(hipEventRecord(w.ev_margA, s));
        DV_CHECK(hipStreamWaitEvent(c0_side, w.ev_margA, 0));
        be_launch_marg_c0(ma, c0_side);
        DV_CHECK(hipGetLastError());
    }
    return 0;
}

// new prior header: kept blocks, indices shifted like addr_shift (estimator.cpp:537-548 / 591-612); x0 = the states the system was linearised at
static void marg_new_prior(const MargPlan& pl, const double* pose, const double* sb, const double* ex, const double* td, double c0, dv_ba_prior* out) {
    std::memset(out, 0, sizeof(*out));
    if (pl.empty) return;
    out->valid = 1; out->n = pl.n; out->c0 = c0;
    int nb = 0;
    auto put = [&](int type, int new_idx, int dim0, int size_local, const double* x0, int gs) {
        dv_ba_prior_block& pb = out->blocks[nb];
        pb.type = type; pb.idx = new_idx; pb.off = dim0 - pl.m; pb.size_local = size_local;
        for (int k = 0; k < gs; ++k) out->x0[nb][k] = x0[k];
        ++nb;
    };
    auto shift = [&](int k) { return pl.mode == 0 ? k - 1 : (k == BE_WIN ? BE_WIN - 1 : k); };
    for (int k = 0; k < BE_NF; ++k) if (pl.pose_dim[k] >= pl.m) put(0, shift(k), pl.pose_dim[k], 6, pose + 7 * k, 7);
    for (int k = 0; k < BE_NF; ++k) if (pl.sb_dim[k] >= pl.m) put(1, shift(k), pl.sb_dim[k], 9, sb + 9 * k, 9);
    for (int c = 0; c < 2; ++c) if (pl.ex_dim[c] >= 0) put(2, c, pl.ex_dim[c], 6, ex + 7 * c, 7);
    if (pl.td_dim >= 0) put(3, 0, pl.td_dim, 1, td, 1);
    out->nblocks = nb;
}

// `slots` trust-region iterations.  speculative: the candidate of every slot but the last is linearised in full (evaluation + reduce into
// the other set) and judged by the next solve kernel; the last one gets the cost-only evaluation and the accept kernel.  The classic form
// (spare slots after a failed / invalid step) spends 5 launches per slot and needs no look-ahead.
// diagnostics (dv_debug_set "hash_log"): a deterministic hash of a device byte range — 256 threads hash interleaved 8-byte words with FNV-1a, thread 0 folds the 256 results in order
__global__ __launch_bounds__(256) void be_dbg_hash_kernel(const unsigned long long* __restrict__ p, size_t words, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[256];
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = threadIdx.x; i < words; i += 256) { h ^= p[i]; h *= 1099511628211ull; }
    sh[threadIdx.x] = h;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 1469598103934665603ull; for (int k = 0; k < 256; ++k) { t ^= sh[k]; t *= 1099511628211ull; } *out = t; }
}
struct BeDbgJob { const unsigned long long* p[BeWork::DBG_RANGES]; unsigned long long words[BeWork::DBG_RANGES]; unsigned long long* out; };
__global__ __launch_bounds__(256) void be_dbg_hash_multi_kernel(BeDbgJob j) {      // blockIdx.x = range
    __shared__ unsigned long long sh[256];
    const unsigned long long* p = j.p[blockIdx.x]; const unsigned long long words = j.words[blockIdx.x];
    unsigned long long h = 1469598103934665603ull;
    for (unsigned long long i = threadIdx.x; i < words; i += 256) { h ^= p[i]; h *= 1099511628211ull; }
    sh[threadIdx.x] = h;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long t = 1469598103934665603ull; for (int k = 0; k < 256; ++k) { t ^= sh[k]; t *= 1099511628211ull; } j.out[blockIdx.x] = t; }
}
// everything a launch of the round may write, hashed behind it on the same stream: 0 packets[0] 1 packets[1] 2 imu_out[0] 3 imu_out[1] 4 prior_out[0] 5 prior_out[1] 6 cand_cost
// 7 Hd[0] 8 Hd[1] 9 Sc[0] 10 Sc[1] 11 gvec[0] 12 gvec[1] 13 x 14 cand 15 ctl
static void be_dbg_stage(dv_ctx* c, int it, int kind, hipStream_t s) {
    BeWork& w = c->be;
    if (!w.debug_hash_log || it >= BeWork::DBG_SLOTS) return;
    const bool light = w.debug_hash_light;      // "hash_light": only the small buffers (a few KB: microsecond kernels), and only behind solve / candidate evaluation — the full form changes the timing so much that the defect does not occur
    if (light && kind != 2 && kind != 3) return;
    if (!w.dbg_slots) { if (hipHostMalloc((void**)&w.dbg_slots, sizeof(unsigned long long) * BeWork::DBG_SLOTS * 5 * BeWork::DBG_RANGES, hipHostMallocDefault) != hipSuccess) return; std::memset(w.dbg_slots, 0, sizeof(unsigned long long) * BeWork::DBG_SLOTS * 5 * BeWork::DBG_RANGES); }
    const size_t n = BE_MAX_STATE, pk = (size_t)BE_PK_SIZE * BE_PK_STRIDE, sb = w.pend->state_bytes / 8;
    BeDbgJob j{};
    const void* ptr[BeWork::DBG_RANGES] = { w.packets[0], w.packets[1], w.imu_out[0], w.imu_out[1], w.prior_out[0], w.prior_out[1], w.cand_cost, w.Hd[0], w.Hd[1], w.Sc[0], w.Sc[1], w.gvec[0], w.gvec[1], w.x, w.cand, w.ctl };
    const size_t io = (size_t)BE_WIN * IMU_OUT_STRIDE, po = (size_t)BE_MAX_PRIOR + 1;
    const size_t words[BeWork::DBG_RANGES] = { pk, pk, io, io, po, po, (size_t)BE_MAX_LM + BE_WIN + 1, n * n, n * n, n * n, n * n, 2 * n, 2 * n, sb, sb, sizeof(BeCtl) / 8 };
    for (int r = 0; r < BeWork::DBG_RANGES; ++r) { j.p[r] = (const unsigned long long*)ptr[r]; j.words[r] = (light && !(r == 6 || r == 11 || r == 12 || r >= 13)) ? 0 : words[r]; }
    j.out = w.dbg_slots + (size_t)(it * 5 + kind) * BeWork::DBG_RANGES;
    hipLaunchKernelGGL(be_dbg_hash_multi_kernel, dim3(BeWork::DBG_RANGES), dim3(256), 0, s, j);
}

static void be_dbg_hash(const void* dev, size_t bytes, unsigned long long* out_pinned, hipStream_t s) {
    if (!dev || bytes < 8) { *out_pinned = 0; return; }
    hipLaunchKernelGGL(be_dbg_hash_kernel, dim3(1), dim3(256), 0, s, (const unsigned long long*)dev, bytes / 8, out_pinned);
}

static int be_enqueue_slots(dv_ctx* ctx, BePending& pd, int slots, bool speculative, hipStream_t s) {
    const bool kt = ctx->timing && ctx->kernel_timing;       // per-launch events (roofline measurement); off in the throughput run
    auto eval = [&](int mode) {
        if (kt) { StageScope k(ctx, mode == BE_EVAL_CAND_COST ? "k_be_eval_cost" : "k_be_eval_full", s); be_launch_eval(pd.ea, mode, s); }
        else be_launch_eval(pd.ea, mode, s);
        be_launch_eval_ext(pd.ea, pd.xt, mode, s);      // (free extrinsic / td blocks only)
    };
    auto reduce = [&](int spec) {
        if (kt) { StageScope k(ctx, "k_be_reduce", s); be_launch_reduce(pd.sa, spec, s); }
        else be_launch_reduce(pd.sa, spec, s);
        be_launch_reduce_ext(pd.sa, spec, s);
    };
    auto solve = [&](int spec) {
        if (kt) { StageScope k(ctx, "k_be_solve", s); return be_launch_solve(pd.sa, spec, s); }
        return be_launch_solve(pd.sa, spec, s);
    };
    auto accept = [&]() {
        if (kt) { StageScope k(ctx, "k_be_accept", s); be_launch_accept(pd.sa, s); }
        else be_launch_accept(pd.sa, s);
    };
    // sharded window (be_shard.hip): every reduce is followed by the exchange of the partial systems and their rank-ordered sum, every cost-only
    // evaluation by the exchange of the candidate costs.  The exchanges are enqueued unconditionally (all ranks hold identical control blocks).
    const bool sharded = pd.sa.sh.on != 0;
    auto exchange_system = [&](int spec) -> int {
        if (!sharded) return 0;
        if (kt) { StageScope k(ctx, "k_be_exchange", s); if (be_exchange(ctx, (size_t)pd.sa.sh.len, s)) return -1; }
        else if (be_exchange(ctx, (size_t)pd.sa.sh.len, s)) return -1;
        if (kt) { StageScope k(ctx, "k_be_shard_finalize", s); be_launch_shard_finalize(pd.sa, spec, s); }
        else be_launch_shard_finalize(pd.sa, spec, s);
        return 0;
    };
    auto exchange_cost = [&]() -> int {
        if (!sharded) return 0;
        be_launch_shard_cost(pd.sa, 0, s);
        if (be_exchange(ctx, 8, s)) return -1;          // one partial sum per rank (padded to 64 bytes)
        be_launch_shard_cost(pd.sa, 1, s);
        return 0;
    };
    for (int it = 0; it < slots; ++it) {
        const bool head = !speculative || it == 0, last = !speculative || it == slots - 1;
        if (head) { eval(BE_EVAL_X); be_dbg_stage(ctx, it, 0, s); reduce(0); be_dbg_stage(ctx, it, 1, s); if (exchange_system(0)) return -1; }
        if (solve(head ? 0 : 1)) DV_FAIL("dv_ba_solve: cannot set dynamic LDS size");
        be_dbg_stage(ctx, it, 2, s);
        if (last) { eval(BE_EVAL_CAND_COST); be_dbg_stage(ctx, it, 3, s); if (exchange_cost()) return -1; if (!(pd.fuse_accept_gauge && !kt && it == slots - 1)) accept(); }      // (fused: be_enqueue_tail launches accept + gauge as one kernel)
        else { eval(BE_EVAL_CAND_FULL); be_dbg_stage(ctx, it, 3, s); reduce(1); be_dbg_stage(ctx, it, 4, s); if (exchange_system(1)) return -1; }
    }
    if (sharded && (!speculative || slots > 0)) {
        // every rank has moved its own landmarks only: one gather of the inverse depths of x behind the pass's last accept decision (whatever reads the whole state —
        // the gauge kernel's download, the outlier test, the marginalization, a spare-slot pass — comes behind it on the stream)
        be_launch_shard_depth(pd.sa, 0, s);
        if (be_exchange(ctx, (size_t)pd.sa.sh.cap, s)) return -1;
        be_launch_shard_depth(pd.sa, 1, s);
    }
    return 0;
}

// gauge fix + marginalization + download, enqueued behind the slots on the same stream (no host round trip)
// gauge fix + download of the states (event) + marginalization, enqueued behind the slots on the same stream: the host
// waits only for the event, so the marginalization of frame k overlaps the host's turnaround and the upload of frame k+1;
// whatever reads its result (the next solve) is ordered behind it on the stream.
// the gauge kernel's arguments of an estimator solve: it writes the gauge-fixed copy, the control block and (dynamic mode: body.para_pose as ceres leaves it, before
// Double2vector's gauge fix) the raw poses straight into the pinned buffer
static void be_gauge_args(dv_ctx* ctx, const BePending& pd, BeGaugeArgs& ga) {
    BeWork& w = ctx->be;
    uint8_t* hp = (uint8_t*)w.pinned;
    ga = BeGaugeArgs{};
    ga.x = w.x; ga.out = w.cand; ga.nlm = pd.nlm; ga.nframes = pd.nframes; ga.use_imu = pd.use_imu;
    std::memcpy(ga.R0, pd.gauge_R0, sizeof(ga.R0)); std::memcpy(ga.ypr0, pd.gauge_ypr0, sizeof(ga.ypr0)); std::memcpy(ga.P0, pd.gauge_P0, sizeof(ga.P0));
    ga.h_out = (BeState*)(hp + w.dl_off); ga.h_ctl = (BeCtl*)(hp + w.dl_off + sizeof(BeState)); ga.ctl = w.ctl; ga.state_doubles = (int)((pd.state_bytes + 7) / 8);
    ga.h_raw_pose = pd.want_raw_pose ? (double*)(hp + w.dl_off + sizeof(BeState) + sizeof(BeCtl) + 256) : nullptr;
}
static int be_enqueue_tail(dv_ctx* ctx, BePending& pd, hipStream_t s) {
    BeWork& w = ctx->be;
    uint8_t* hp = (uint8_t*)w.pinned;
    pd.ev_state_ext = nullptr;                                  // (a tail of its own: the host waits for w.ev_state again)
    BeGaugeArgs ga{};
    BeState* hx = (BeState*)(hp + w.dl_off); BeCtl* hctl = (BeCtl*)(hp + w.dl_off + sizeof(BeState));
    static_assert(sizeof(BeCtl) % sizeof(double) == 0 && sizeof(BeState) % sizeof(double) == 0, "downloaded as doubles");
    if (pd.fused_present) {        // estimator path
        be_gauge_args(ctx, pd, ga);
        const bool kt = ctx->timing && ctx->kernel_timing;
        if (pd.fuse_accept_gauge && !kt) be_launch_accept_gauge(pd.sa, ga, s); else be_launch_gauge(ga, s);
        DV_CHECK(hipGetLastError());
    } else {
        DV_CHECK(hipMemcpyAsync(hx, w.x, pd.state_bytes, hipMemcpyDeviceToHost, s));
        DV_CHECK(hipMemcpyAsync(hctl, w.ctl, sizeof(BeCtl), hipMemcpyDeviceToHost, s));
    }
    if (pd.rej_on) { be_launch_reject(pd.rej, s); DV_CHECK(hipGetLastError()); }      // reads the gauge-fixed copy (w.cand), writes its flags to pinned memory: part of what ev_state covers
    DV_CHECK(hipEventRecord(w.ev_state, s));
    if (pd.fused_present && pd.do_marg && !pd.pl.empty) {
        const bool side = w.c0_side && !(ctx->timing && ctx->kernel_timing);      // (the per-kernel timing mode keeps the whole marginalization on the BA stream: k_be_marg)
        if (side && !w.c0_stream) {
            DV_CHECK(hipStreamCreateWithFlags(&w.c0_stream, hipStreamNonBlocking));
            DV_CHECK(hipEventCreateWithFlags(&w.ev_margA, hipEventDisableTiming)); DV_CHECK(hipEventCreateWithFlags(&w.ev_c0, hipEventDisableTiming));
        }
        hipStream_t cs = side ? w.c0_stream : s;
        if (marg_enqueue(ctx, pd.pl, w.cand, pd.g_norm, w.priorA, w.priorb, w.priorA_buf[pd.nxt], w.priorb_buf[pd.nxt], w.marg_scal, w.prior_c0 + pd.nxt, s, side ? w.c0_stream : nullptr)) return -1;
        double* hscal = (double*)(hp + w.dl_off + sizeof(BeState) + sizeof(BeCtl)) + 4 * pd.scal_slot;      // two alternating host slots
        DV_CHECK(dv_copy_async(hscal, w.marg_scal, 32, cs));
        if (side) { DV_CHECK(hipEventRecord(w.ev_c0, cs)); w.c0_pending = true; }
        pd.marg_in_flight = true;
    }
    return 0;
}

// the marginalization enqueued by the PREVIOUS frame reports its health here (its 4 scalars were downloaded behind it)
static int be_check_prev_marg(dv_ctx* ctx, BePending& pd) {
    if (!pd.marg_check_due) return 0;
    pd.marg_check_due = false;
    const double* hscal = (const double*)((uint8_t*)ctx->be.pinned + ctx->be.dl_off + sizeof(BeState) + sizeof(BeCtl)) + 4 * pd.check_slot;
    // hscal[2] != 0: a pivot of A_mm was <= 1e-8 and was skipped on the device (pseudo-inverse, as the reference's eigen clamp does,
    // marginalization_factor.cpp:286-289).  The prior stays finite and usable, so the frame is never aborted half-way; the event is only counted.
    if (hscal[2] != 0.0) ctx->be.marg_clamped++;
    std::memcpy(ctx->be.marg_last, hscal, 32); ctx->be.marg_checked++;
    return 0;
}

static int be_begin_impl(dv_ctx* ctx, dv_ba_problem* P, BeFused* fused, bool eval_only) {
    if (!ctx) return -1;
    BePending& pd = *ctx->be.pend;
    if (pd.active) DV_FAIL("dv_ba_solve: previous solve not collected");
    pd.trivial = false;
    const std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    if (!P || !P->pose || !P->ex_pose || !P->td) DV_FAIL("dv_ba_solve: null argument");
    if (P->nframes < 1 || P->nframes > BE_NF) DV_FAIL("dv_ba_solve: nframes out of range");
    if (P->nlm < 0 || P->nlm > BE_MAX_LM) DV_FAIL("dv_ba_solve: more than kNumFeat=1000 landmarks");
    if (P->nimu < 0 || P->nimu > BE_WIN) DV_FAIL("dv_ba_solve: bad IMU factor count");
    if (P->use_imu && !P->speed_bias) DV_FAIL("dv_ba_solve: speed_bias is null");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (be_ensure(ctx, P->nfac)) return -1;
    BeWork& w = ctx->be;
    hipStream_t s = ctx->be_stream;
    // ---- the bulk of the upload FIRST: landmark and factor tables (nine tenths of the bytes) are complete when the call begins (the estimator builds the factors in the pinned
    // mirror itself), so their staging copy — 15 of the 19 us the whole upload takes over PCIe, on the path between two frames' solves — runs while the host still works out the
    // column layout, the IMU records and the marginalization plan below; the head of the region (control block, states, IMU, prior header, index tables) follows at the end.
    // Measured (profiles/r06_experiments/upload_split_ab.json, 100-step blocks interleaved on one box): +0.1 - 0.7 %, i.e. noise level — the BA stream is still busy with the
    // previous frame's marginalization when the early copy arrives; kept because it cannot lose ----
    static const bool split_upload = [] { const char* e = std::getenv("DVINS_UPLOAD_SPLIT"); return !(e && e[0] == '0'); }();      // 0: one copy at the end (rounds 1 - 5; A/B)
    {
        uint8_t* hp0 = (uint8_t*)w.pinned;
        if (P->nlm) std::memcpy(hp0 + w.up_lm, P->landmarks, sizeof(BeLm) * (size_t)P->nlm);
        if (P->nfac && (const void*)P->factors != (const void*)(hp0 + w.up_fac)) std::memcpy(hp0 + w.up_fac, P->factors, sizeof(BeFactor) * (size_t)P->nfac);      // the estimator builds the table in place
        if (split_upload) DV_CHECK(dv_copy_async((uint8_t*)w.block.p + w.up_lm, hp0 + w.up_lm, (w.up_fac - w.up_lm) + sizeof(BeFactor) * (size_t)P->nfac, s));
    }
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
    // free extrinsic / td blocks (estimator.cpp:87-100; dv_ba_problem::free_blocks): their columns follow the frames' (ceres' reduced program keeps the insertion order of
    // AddBodyParameterBlock: poses and speed-biases, the extrinsics, td); ext entry q = 0..5 ex0, 6..11 ex1, 12 td
    BeExt xt{};
    for (int q = 0; q <= BE_NX; ++q) xt.xcol[q] = -1;
    if (P->free_blocks & ~3) DV_FAIL("dv_ba_solve: unknown bits in free_blocks");
    if (P->free_blocks & 1) for (int cidx = 0; cidx < 2; ++cidx) for (int k = 0; k < 6; ++k) { xt.xcol[6 * cidx + k] = col; col_kind[col] = 2; col_frame[col] = 100 + cidx; col_comp[col] = k; ++col; }
    if (P->free_blocks & 2) { xt.xcol[12] = col; col_kind[col] = 3; col_frame[col] = 102; col_comp[col] = 0; ++col; }
    xt.on = (P->free_blocks & 3) ? 1 : 0;
    d.nstate = col;
    d.pad = 0;
    const bool has_prior = P->prior && P->prior->valid;
    const bool prior_on_device = has_prior && w.prior_resident && P->prior_A == w.priorA_buf[w.prior_cur];
    if (col == 0) {       // e.g. vision-only, first frame: pose 0 is constant and no landmark has 4 observations yet
        if (P->nlm > 0) DV_FAIL("dv_ba_solve: landmarks without a free pose block");
        if (fused && fused->marg_mode >= 0) DV_FAIL("dv_ba_solve: marginalization needs a full window");
        pd.active = true; pd.trivial = true;
        return 0;
    }
    BePriorHdr ph{};
    if (has_prior) {
        std::memcpy(&ph, P->prior, sizeof(ph));
        if (ph.n > BE_MAX_PRIOR || ph.nblocks > 16) DV_FAIL("dv_ba_solve: prior too large");
        for (int b = 0; b < ph.nblocks; ++b) {
            const BePriorBlock pb = ph.blocks[b];
            int c0 = -1;
            if (pb.type == 0) c0 = d.pose_col[pb.idx]; else if (pb.type == 1) c0 = d.sb_col[pb.idx];
            else if (pb.type == 2) c0 = xt.xcol[6 * pb.idx]; else if (pb.type == 3) c0 = xt.xcol[12];
            if (c0 >= 0) for (int k = 0; k < pb.size_local; ++k) prior_col[c0 + k] = pb.off + k;
        }
    }
    // ---- marginalization structure (does not depend on the solution): planned now so that its tables ride in the same upload ----
    MargPlan& pl = pd.pl;
    const bool do_marg = fused && fused->marg_mode >= 0;
    if (do_marg) {
        if (P->nframes != BE_NF) DV_FAIL("dv_marginalize: needs a full window (frame == kWinSize)");
        std::vector<int> sel;
        if (fused->marg_mode == 0) for (int l = 0; l < P->nlm; ++l) if (P->landmarks[l].anchor == 0) sel.push_back(l);
        const bool imu01 = P->nimu > 0 && P->imu[0].fi == 0 && P->imu[0].fj == 1;
        if (marg_plan(ctx, pl, fused->marg_mode, has_prior ? P->prior : nullptr, P->factors, P->landmarks, sel.data(), (int)sel.size(), imu01)) return -1;
    }
    // ---- upload: everything is staged in the pinned mirror of the device's upload region and travels in ONE copy ----
    uint8_t* hp = (uint8_t*)w.pinned;
    BeState* hx = (BeState*)(hp + w.up_x);
    std::memset(hx, 0, offsetof(BeState, inv_depth));
    for (int f = 0; f < P->nframes; ++f) { std::memcpy(hx->pose[f], P->pose + 7 * f, 56); if (P->use_imu) std::memcpy(hx->sb[f], P->speed_bias + 9 * f, 72); }
    std::memcpy(hx->ex, P->ex_pose, 14 * 8); hx->td = P->td[0];
    if (P->nlm) std::memcpy(hx->inv_depth, P->inv_depth, 8 * (size_t)P->nlm);
    const size_t state_bytes = offsetof(BeState, inv_depth) + 8 * (size_t)P->nlm;
    BeCtl* hctl = (BeCtl*)(hp + w.up_ctl);
    std::memset(hctl, 0, sizeof(BeCtl));
    hctl->need_eval = 1; hctl->first = 1; hctl->max_iters = P->max_iters; hctl->radius = 1e4; hctl->mu = eval_only ? 0.0 : 1e-8; hctl->step_valid = 0;
    BeImu* himu = (BeImu*)(hp + w.up_imu);
    for (int k = 0; k < P->nimu; ++k) {
        const double* hint = (k < (int)w.sqrt_hint.size()) ? w.sqrt_hint[k] : nullptr;      // the estimator caches U per pre-integration (Q8)
        if (be_fill_imu(P->imu[k], himu[k], hint)) DV_FAIL("dv_ba_solve: IMU covariance is singular");
    }
    std::memcpy(hp + w.up_prior, &ph, sizeof(ph));
    std::memcpy(hp + w.up_idx, idx.data(), 4 * idx.size());
    if (do_marg && !pl.empty) std::memcpy(hp + w.up_mt, pl.tab, sizeof(pl.tab));
    DV_CHECK(dv_copy_async(w.block.p, hp, split_upload ? w.up_lm : w.up_fac + sizeof(BeFactor) * (size_t)P->nfac, s));      // the head of the upload region (the landmark / factor tables went first, see the top); kernels reading the pinned mirror: copy.hip
    if (w.c0_pending) { DV_CHECK(hipStreamWaitEvent(s, w.ev_c0, 0)); w.c0_pending = false; }      // the previous frame's c0 (side stream) and its health scalars: before anything reads the prior's constant
    if (has_prior && !prior_on_device) {               // a prior handed over in host memory (the estimator's stays in HBM)
        if (!P->prior_A || !P->prior_b) DV_FAIL("dv_ba_solve: prior without A / b");
        DV_CHECK(hipMemcpyAsync(w.priorA_buf[w.prior_cur], P->prior_A, 8 * (size_t)ph.n * ph.n, hipMemcpyHostToDevice, s));
        DV_CHECK(hipMemcpyAsync(w.priorb_buf[w.prior_cur], P->prior_b, 8 * (size_t)ph.n, hipMemcpyHostToDevice, s));
        double* hc0 = (double*)(hp + w.dl_off + sizeof(BeState) + sizeof(BeCtl) + 128); *hc0 = ph.c0;
        DV_CHECK(hipMemcpyAsync(w.prior_c0 + w.prior_cur, hc0, 8, hipMemcpyHostToDevice, s));
        w.prior_resident = false;
    }
    w.priorA = w.priorA_buf[w.prior_cur]; w.priorb = w.priorb_buf[w.prior_cur];
    std::chrono::steady_clock::time_point t_up = std::chrono::steady_clock::now();
    // ---- schedule ----
    BeEvalArgs ea{};
    ea.ctl = w.ctl; ea.x = w.x; ea.cand = w.cand; ea.fac = w.fac; ea.lm = w.lm; ea.imu = w.imu; ea.prior = w.prior; ea.priorA = w.priorA; ea.priorb = w.priorb;
    ea.dims = d; ea.g_norm = P->g_norm; ea.cand_cost = w.cand_cost; ea.lm_obs = w.lm_obs; ea.prior_c0 = w.prior_c0 + w.prior_cur;
    ea.lm_lo = 0; ea.lm_hi = P->nlm;
    if (P->free_blocks & 3) {
        if (ctx->dist.transport != 0) DV_FAIL("dv_ba_solve: free extrinsic / td blocks are not built for the landmark-sharded window");
        if (xt.on) {
            DV_CHECK(w.xpk_buf.ensure(2 * 8 * (size_t)BX_SIZE * BE_PK_STRIDE));
            xt.xpk[0] = (double*)w.xpk_buf.p; xt.xpk[1] = xt.xpk[0] + (size_t)BX_SIZE * BE_PK_STRIDE;
        }
    }
    BeShard sh{};
    if (ctx->dist.transport != 0) {        // landmark-sharded window: contiguous ranges of cap = ceil(nlm / world) landmarks
        const DvDist& dd = ctx->dist;
        sh.on = 1; sh.rank = dd.rank; sh.world = dd.world;
        sh.cap = std::max(1, (P->nlm + dd.world - 1) / dd.world);
        sh.lo = std::min(P->nlm, dd.rank * sh.cap); sh.hi = std::min(P->nlm, sh.lo + sh.cap);
        sh.len = BE_XS_LEN;
        sh.xsend = (double*)dd.xsend.p; sh.xrecv = (const double*)dd.xrecv.p;
        sh.qf[0] = (double*)dd.qf.p; sh.qf[1] = sh.qf[0] + BE_QF_LEN;
        ea.lm_lo = sh.lo; ea.lm_hi = sh.hi;
    }
    for (int k = 0; k < 2; ++k) { ea.packets[k] = w.packets[k]; ea.imu_out[k] = w.imu_out[k]; ea.prior_out[k] = w.prior_out[k]; }
    BeSolveArgs sa{};
    sa.ctl = w.ctl; sa.x = w.x; sa.cand = w.cand; sa.lm = w.lm; sa.imu = w.imu; sa.prior = w.prior; sa.priorA = w.priorA; sa.dims = d;
    sa.cand_cost = w.cand_cost; sa.lm_obs = w.lm_obs;
    for (int k = 0; k < 2; ++k) { sa.packets[k] = w.packets[k]; sa.imu_out[k] = w.imu_out[k]; sa.prior_out[k] = w.prior_out[k]; sa.Hd[k] = w.Hd[k]; sa.Sc[k] = w.Sc[k]; sa.gvec[k] = w.gvec[k]; }
    sa.scale_p = w.scale_p; sa.diag_p = w.diag_p; sa.grad_p = w.grad_p; sa.gn_p = w.gn_p; sa.scale_l = w.scale_l; sa.diag_l = w.diag_l; sa.grad_l = w.grad_l; sa.gn_l = w.gn_l;
    sa.prior_col = w.prior_col; sa.col_kind = w.col_kind; sa.col_frame = w.col_frame; sa.col_comp = w.col_comp;
    sa.xnorm2_extra = P->x_norm2_extra; sa.sh = sh; sa.xt = xt;
    // the 16-wide MFMA factorisation (be_solve.hip MF16) wherever the system fits its tile budget (n <= 175: every window the estimator builds); "ldl_generic" selects the 4-wide panel form
    sa.ldl_mf16 = 0;
    if (!w.ldl_generic && !(P->free_blocks & 3)) {      // (free extrinsic / td blocks: the generic form carries the ext entries)
        uint8_t plan[64];
        if (be_mf16_plan(d.nstate, plan)) { std::memcpy(sa.ldl_col0, plan, sizeof(plan)); sa.ldl_mf16 = 1; }
    }
    // The first pass enqueues exactly max_iters slots: enough unless a linear solve failed (mu *= 10 retry) or a step was
    // invalid; be_solve_fused_end checks the downloaded control block and, in that rare case, runs the spare slots and the
    // (idempotent) tail again.
    pd.rej_on = false;
    if (fused && fused->want_reject && w.gpu_reject && P->nlm > 0) {
        if (!w.rej_pinned) DV_CHECK(hipHostMalloc((void**)&w.rej_pinned, BE_MAX_LM, hipHostMallocDefault));
        BeRejectArgs& r = pd.rej;
        r.st = w.cand; r.fac = w.fac; r.lm = w.lm; r.nlm = P->nlm; r.nframes = P->nframes; r.focal = fused->rej_focal; r.flags = w.rej_pinned;
        std::memcpy(r.ric, fused->rej_ric, sizeof(r.ric)); std::memcpy(r.tic, fused->rej_tic, sizeof(r.tic));
        r.ex_from_state = (P->free_blocks & 1) ? 1 : 0; r.pad = 0;
        pd.rej_on = true;
    }
    pd.xt = xt; pd.copy_ex_td = (P->free_blocks & 3) != 0;
    pd.ea = ea; pd.sa = sa; pd.fused_present = fused != nullptr; pd.fuse_accept_gauge = fused != nullptr && !ctx->batch && !sh.on /* sharded: the last accept decision must stand before the gather of the inverse depths */; pd.max_iters = P->max_iters; pd.g_norm = P->g_norm; pd.nframes = P->nframes; pd.use_imu = P->use_imu; pd.nlm = P->nlm;
    pd.want_raw_pose = fused && fused->want_raw_pose;
    if (fused) { std::memcpy(pd.gauge_R0, fused->R0, sizeof(pd.gauge_R0)); std::memcpy(pd.gauge_ypr0, fused->ypr0, sizeof(pd.gauge_ypr0)); std::memcpy(pd.gauge_P0, fused->P0, sizeof(pd.gauge_P0)); }
    pd.do_marg = do_marg; pd.state_bytes = state_bytes; pd.nxt = 1 - w.prior_cur;
    if (eval_only) {        // dv_ba_eval: one evaluation + assembly of the reduced camera system at the given states (mu = 0)
        be_launch_eval(ea, BE_EVAL_X, s); be_launch_eval_ext(ea, xt, BE_EVAL_X, s);
        be_launch_reduce(sa, 0, s); be_launch_reduce_ext(sa, 0, s);
        if (sh.on) { if (be_exchange(ctx, (size_t)sh.len, s)) return -1; be_launch_shard_finalize(sa, 0, s); }
        DV_CHECK(hipGetLastError());
        return 0;
    }
    if (w.debug_hash_log && fused) {        // what the device holds when the round starts: the uploaded block as it arrived, the prior it will read
        if (!w.dbg_pinned) DV_CHECK(hipHostMalloc((void**)&w.dbg_pinned, 64, hipHostMallocDefault));
        const size_t up_bytes = w.up_fac + sizeof(BeFactor) * (size_t)P->nfac;
        be_dbg_hash(w.block.p, up_bytes & ~(size_t)7, w.dbg_pinned + 0, s);
        const bool pv = P->prior && P->prior->valid;
        if (pv) { be_dbg_hash(w.priorA, 8 * (size_t)P->prior->n * P->prior->n, w.dbg_pinned + 1, s); be_dbg_hash(w.priorb, 8 * (size_t)P->prior->n, w.dbg_pinned + 2, s); } else { w.dbg_pinned[1] = 0; w.dbg_pinned[2] = 0; }
    }
    const int first_slots = w.debug_short_first_pass ? std::max(1, P->max_iters - 2) : P->max_iters;      // dv_debug_set(ctx, "short_first_pass", 1): tests exercise the spare-slot path
    if (ctx->batch && fused && !sh.on) {       // member of a dv_batch: the upload is on its way; dv_batch_enqueue launches the slots of all members together
        pd.deferred = true; pd.first_slots = first_slots;
        pd.active = true; pd.t_begin = t_begin; pd.t_up = t_up; pd.t_enq = t_up;
        return 0;
    }
    { StageScope sc(ctx, "ba_solve", s); if (be_enqueue_slots(ctx, pd, first_slots, true, s)) return -1; }
    std::chrono::steady_clock::time_point t_enq = std::chrono::steady_clock::now();
    if (be_enqueue_tail(ctx, pd, s)) return -1;
    pd.active = true; pd.t_begin = t_begin; pd.t_up = t_up; pd.t_enq = t_enq;
    return 0;
}

void* be_staging_factors(dv_ctx* ctx, int* cap) {
    if (!ctx || ctx->be.pend->active) return nullptr;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess || be_ensure(ctx, 0)) return nullptr;
    *cap = ctx->be.fac_cap;
    return (uint8_t*)ctx->be.pinned + ctx->be.up_fac;
}

int be_solve_fused_begin(dv_ctx* ctx, dv_ba_problem* P, BeFused* fused) { return be_begin_impl(ctx, P, fused, false); }

int be_solve_fused_end(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary, BeFused* fused) {
    if (!ctx) return -1;
    BePending& pd = *ctx->be.pend;
    if (!pd.active) DV_FAIL("dv_ba_solve: nothing to collect");
    pd.active = false;
    if (pd.trivial) {
        if (summary) { summary->iterations = 0; summary->successful = 0; summary->termination = 1; summary->slots = 0; summary->initial_cost = 0; summary->final_cost = 0; }
        return 0;
    }
    BeWork& w = ctx->be;
    hipStream_t s = ctx->be_stream;
    if (pd.deferred) {        // a batch member collected without dv_batch_enqueue (the synchronous solves of the initialisation): enqueue it alone
        pd.deferred = false;
        if (be_enqueue_slots(ctx, pd, pd.first_slots, true, s)) return -1;
        if (be_enqueue_tail(ctx, pd, s)) return -1;
    }
    const MargPlan& pl = pd.pl;
    uint8_t* hp = (uint8_t*)w.pinned;
    const BeState* hx = (const BeState*)(hp + w.dl_off); const BeCtl* hctl = (const BeCtl*)(hp + w.dl_off + sizeof(BeState));
    DV_CHECK(hipEventSynchronize(pd.ev_state_ext ? pd.ev_state_ext : w.ev_state));
    if (w.debug_wait_tail) DV_CHECK(hipStreamSynchronize(ctx->be_stream));      // dv_debug_set "wait_tail": the host does not move on until the marginalization behind ev_state has drained too (bisecting the open multi-sequence defect)      // (member of a dv_batch round: the group's event behind the shared gauge / reject launches)
    if (be_dist_check(ctx)) return -1;      // sharded window, peer transport: a dead or late peer is an error of THIS solve, not garbage in its result
    // the previous frame's marginalization ran before this frame's upload (stream order), so its scalars have landed
    if (be_check_prev_marg(ctx, pd)) return -1;
    if (!hctl->done) {        // rare: a failed linear solve / invalid step used up slots -> the 3 spare slots, then the tail once more
        // (the raw solution is still in w.x: the gauge fix writes to the candidate buffer; the marginalization reads the untouched old prior)
        if (be_enqueue_slots(ctx, pd, 3, false, s)) return -1;
        if (be_enqueue_tail(ctx, pd, s)) return -1;
        DV_CHECK(hipEventSynchronize(w.ev_state));
        if (be_dist_check(ctx)) return -1;
    }
    if (pd.marg_in_flight) { pd.marg_in_flight = false; pd.marg_check_due = true; pd.check_slot = pd.scal_slot; pd.scal_slot ^= 1; }
    if (ctx->host_timing) {
        StageTimer* te = dv_timer_for(ctx, "h_solve_enqueue"); te->total_ms += std::chrono::duration<double, std::milli>(pd.t_enq - pd.t_up).count(); te->count++;
        StageTimer* tu = dv_timer_for(ctx, "h_solve_upload"); tu->total_ms += std::chrono::duration<double, std::milli>(pd.t_up - pd.t_begin).count(); tu->count++;
    }
    if (ctx->timing) {
        DV_CHECK(hipStreamSynchronize(s));         // measurement mode only: the timers are harvested from an idle stream
        StageTimer* te = dv_timer_for(ctx, "h_solve_enqueue"); te->total_ms += std::chrono::duration<double, std::milli>(pd.t_enq - pd.t_up).count(); te->count++;
        StageTimer* tu = dv_timer_for(ctx, "h_solve_upload"); tu->total_ms += std::chrono::duration<double, std::milli>(pd.t_up - pd.t_begin).count(); tu->count++;
        dv_harvest_timers(ctx, s);
    }
    for (int f = 0; f < P->nframes; ++f) { std::memcpy(P->pose + 7 * f, hx->pose[f], 56); if (P->use_imu) std::memcpy(P->speed_bias + 9 * f, hx->sb[f], 72); }
    if (P->nlm) std::memcpy(P->inv_depth, hx->inv_depth, 8 * (size_t)P->nlm);
    if (pd.copy_ex_td) { std::memcpy(P->ex_pose, hx->ex, 14 * 8); P->td[0] = hx->td; }      // free blocks: para_ex_pose / para_td as the solve left them
    if (w.debug_hash_log && fused && w.dbg_pinned) {      // what the device holds when the round is over: x (raw solution), the candidate buffer (gauge-fixed copy), the control block
        be_dbg_hash(w.x, pd.state_bytes & ~(size_t)7, w.dbg_pinned + 3, s);
        be_dbg_hash(w.cand, pd.state_bytes & ~(size_t)7, w.dbg_pinned + 4, s);
        be_dbg_hash(w.ctl, sizeof(BeCtl) & ~(size_t)7, w.dbg_pinned + 5, s);
        DV_CHECK(hipStreamSynchronize(s));
        w.dbg_dev_log.push_back(w.dbg_solve_no++);
        for (int k = 0; k < 6; ++k) w.dbg_dev_log.push_back(w.dbg_pinned[k]);
        if (w.dbg_slots) {
            const size_t nv = (size_t)BeWork::DBG_SLOTS * 5 * BeWork::DBG_RANGES;
            w.dbg_slot_log.insert(w.dbg_slot_log.end(), w.dbg_slots, w.dbg_slots + nv);
            std::memset(w.dbg_slots, 0, sizeof(unsigned long long) * nv);
        }
    }
    if (summary) {
        summary->iterations = hctl->iter; summary->successful = hctl->successful; summary->termination = hctl->done ? hctl->termination : 0;
        summary->slots = hctl->slots; summary->initial_cost = hctl->initial_cost; summary->final_cost = hctl->x_cost;
    }
    if (fused) fused->rej_flags = pd.rej_on ? w.rej_pinned : nullptr;
    if (fused && pd.want_raw_pose) std::memcpy(fused->raw_pose, hp + w.dl_off + sizeof(BeState) + sizeof(BeCtl) + 256, sizeof(fused->raw_pose));
    if (pd.do_marg) {
        std::memcpy(fused->diag, w.marg_last, sizeof(fused->diag));      // the scalars of THIS frame's marginalization are still in flight: the previous frame's (be_check_prev_marg)
        if (pl.empty) { std::memset(&fused->new_prior, 0, sizeof(fused->new_prior)); w.prior_resident = false; }
        else {
            // header only: A', b' and c0 are (being) written in HBM by the marginalization kernels still in flight
            marg_new_prior(pl, P->pose, P->speed_bias, P->ex_pose, P->td, std::nan(""), &fused->new_prior);      // x0 = the gauge-fixed states just downloaded
            w.prior_cur = pd.nxt; w.prior_resident = true;
            w.priorA = w.priorA_buf[pd.nxt]; w.priorb = w.priorb_buf[pd.nxt];
        }
    }
    return 0;
}

// ---- dv_batch: several independent windows (one estimator each, same device) whose solve slots share every launch -----------------------------------
// Each member keeps its own BA stream for the upload, the gauge fix, the state download and the marginalization; the iteration slots — the launch-bound
// part: 3 launches per iteration and window — run on the batch's stream as ONE launch per stage for all windows (argument tables in HBM, window index in
// the grid).  Per round: S event waits (uploads done), 3 x iterations launches, one event, S event waits (tails).
struct dv_batch {
    std::vector<dv_ctx*> members; int index = 0;          // index: creation order in the process (dv_group_stream_create)
    hipStream_t stream = nullptr; hipEvent_t ev_slots = nullptr;
    hipStream_t solve_stream = nullptr; std::vector<hipEvent_t> ev_x;      // DVINS_SOLVE_CUS: the solve launches of a round on a stream of their own (reserved CUs), chained to `stream` by events
    DevBuf tab; void* tab_pinned = nullptr; size_t tab_bytes = 0;      // [S] BeEvalArgs | [S] BeSolveArgs | [S] BeGaugeArgs | [S] BeRejectArgs | [S] BeMargArgs
    long long batched_rounds = 0, single_rounds = 0;
    DvFrontBatch* front = nullptr;                // the members' front ends in shared launches (dv_batch_track_enqueue, dvins_api.hip)
    hipEvent_t ev_state = nullptr;                // behind the shared accept + gauge + reject launches of a round: what the members' dv_est_process_end wait for
    std::mutex mu; std::condition_variable cv; int arrived = 0; long long generation = 0; int last_rc = 0;      // dv_batch_arrive
    bool aborted = false;                         // dv_batch_abort: every waiting and every later dv_batch_arrive returns -1
    // dv_batch_timing: HIP events around the three launches of the SECOND iteration slot of every round (a steady-state slot: candidate evaluation, reduce, solve with
    // the accept decision), on the batch stream they are launched on; harvested when the next round starts (the events of the previous round have completed by then)
    bool timing = false; hipEvent_t tev[4] = { nullptr, nullptr, nullptr, nullptr }; bool tev_pending = false;
    double t_ms[3] = { 0, 0, 0 }; long long t_n = 0; int t_windows = 0;
};
// dv_destroy of a member: the batch forgets it (a destroyed ctx must never be reached through B->members); threads waiting in dv_batch_arrive
// for a round this member will never join are released with an error
DvFrontBatch*& be_batch_front(dv_batch* B) { return B->front; }
int be_batch_index(dv_batch* B) { return B->index; }
// DVINS_SOLVE_CUS=k (experiment; measured SLOWER and left off: 16 sequences 6.75 -> 3.1 k frames/s, 64 sequences 9.7 -> 6.0 k — the two cross-stream edges per
// iteration slot cost more than the solve's wait for a free CU; profiles/r05_experiments/solve_cus_and_eval_split_ab.txt): k CUs of every XCD are kept for the single-workgroup
// window solves of the dv_batch groups (a be_solve_batch workgroup
// needs a whole CU — 157 KB of LDS — and otherwise waits until one drains of the wide grids' workgroups); every group stream and front-end stream gets the complement.
// CU mask bit i = CU i / 8 of XCD i % 8 (scripts/dbg/cumask_probe.hip).
static int dv_solve_cus() { static int k = -1; if (k < 0) { const char* e = std::getenv("DVINS_SOLVE_CUS"); k = e ? std::atoi(e) : 0; if (k < 0 || k > 16) k = 0; } return k; }
// CAVEAT on the two experiments below: hipExtStreamCreateWithCUMask has no flags argument — its streams are default-flag (blocking) streams that synchronise
// implicitly with the NULL stream, while every other stream of the library is hipStreamNonBlocking; PyTorch's default stream IS the NULL stream.  The "measured slower" of
// DESIGN_HISTORY (round 5) therefore includes whatever that implicit ordering cost in a process that also renders on torch's default stream; CU partitioning is recorded as
// "slower in bench.py", not as rejected in general.
hipError_t dv_solve_stream_create(hipStream_t* s) {
    const int k = dv_solve_cus();
    if (!k) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    uint32_t mask[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int b = 0; b < 8 * k; ++b) mask[b >> 5] |= 1u << (b & 31);
    return hipExtStreamCreateWithCUMask(s, 8, mask);
}
hipError_t dv_group_stream_create(hipStream_t* s, int group_index) {
    if (const int k = dv_solve_cus()) {
        uint32_t mask[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        for (int b = 8 * k; b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
        return hipExtStreamCreateWithCUMask(s, 8, mask);
    }
    const char* e = std::getenv("DVINS_CU_PARTITIONS");
    const int P = e ? std::atoi(e) : 0;
    if (P < 2 || P > 32 || 256 % P) return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    uint32_t mask[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const int per = 256 / P, g = group_index % P;
    for (int b = g * per; b < (g + 1) * per; ++b) mask[b >> 5] |= 1u << (b & 31);
    return hipExtStreamCreateWithCUMask(s, 8, mask);
}
const std::vector<dv_ctx*>& be_batch_members(dv_batch* B) { return B->members; }
void be_batch_detach(dv_ctx* ctx) {
    dv_batch* B = ctx->batch;
    if (!B) return;
    {
        std::lock_guard<std::mutex> lk(B->mu);
        B->members.erase(std::remove(B->members.begin(), B->members.end(), ctx), B->members.end());
        ctx->batch = nullptr;
        if (B->stream) (void)hipStreamSynchronize(B->stream);
        dv_front_batch_sync(B->front);
        if (ctx->be_stream_own) { ctx->be_stream = ctx->be_stream_own; ctx->be_stream_own = nullptr; }
        if (B->arrived > 0) { B->last_rc = -1; B->arrived = 0; ++B->generation; dv_set_error(nullptr, "dv_batch_arrive: a member was destroyed during the round"); }
    }
    B->cv.notify_all();
}
static int batch_enqueue_impl(dv_batch* B) {
    std::vector<dv_ctx*> M;
    for (dv_ctx* c : B->members) if (c->be.pend->active && c->be.pend->deferred && !c->be.pend->trivial) M.push_back(c);
    if (M.empty()) return 0;
    dv_ctx* ctx = M[0];
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = B->stream;                    // == every member's be_stream: their uploads (and the marginalizations of their previous frames) are ordered before the slots
    bool uniform = true; int slots = M[0]->be.pend->first_slots, max_grid = 0, max_n = 0;
    for (dv_ctx* c : M) {
        const BePending& pd = *c->be.pend;
        if (pd.sa.ldl_mf16 == 0 || pd.first_slots != slots || (c->timing && c->kernel_timing) || !pd.fused_present) uniform = false;
        max_grid = std::max(max_grid, be_eval_batch_blocks(pd.ea.dims.nlm, pd.ea.dims.nimu)); max_n = std::max(max_n, pd.sa.dims.nstate);
    }
    if (!uniform || M.size() == 1) {      // mixed kernel variants (or nothing to share): every member's own launches, one member after the other
        for (dv_ctx* c : M) { BePending& pd = *c->be.pend; pd.deferred = false; if (be_enqueue_slots(c, pd, pd.first_slots, true, s) || be_enqueue_tail(c, pd, s)) { dv_set_error(ctx, c->err); return -1; } }
        B->single_rounds++;
        return 0;
    }
    const int S = (int)M.size();
    const size_t cap = B->members.size();
    BeEvalArgs* hea = (BeEvalArgs*)B->tab_pinned; BeSolveArgs* hsa = (BeSolveArgs*)(hea + cap);
    BeGaugeArgs* hga = (BeGaugeArgs*)(hsa + cap); BeRejectArgs* hrj = (BeRejectArgs*)(hga + cap); BeMargArgs* hma = (BeMargArgs*)(hrj + cap);
    // the frame tails' arguments are known now as well (nothing in them depends on the solve): accept + gauge + download, outlier test, marginalization
    int max_rej = 0, max_mlm = 0, any_imu = 0, max_D = 0, n_marg = 0; size_t max_fin = 0;
    for (int i = 0; i < S; ++i) {
        dv_ctx* c = M[i]; BePending& pd = *c->be.pend; BeWork& w = c->be;
        hea[i] = pd.ea; hsa[i] = pd.sa;
        be_gauge_args(c, pd, hga[i]);
        hrj[i] = BeRejectArgs{};
        if (pd.rej_on) { hrj[i] = pd.rej; max_rej = std::max(max_rej, pd.rej.nlm); }
        hma[i] = BeMargArgs{};                    // D = 0: no marginalization for this member this frame
        if (pd.do_marg && !pd.pl.empty) {
            uint8_t* hp = (uint8_t*)w.pinned;
            double* hscal = (double*)(hp + w.dl_off + sizeof(BeState) + sizeof(BeCtl)) + 4 * pd.scal_slot;      // the health scalars go straight to the member's pinned slot
            if (marg_args(c, pd.pl, w.cand, pd.g_norm, w.priorA, w.priorb, w.priorA_buf[pd.nxt], w.priorb_buf[pd.nxt], hscal, w.prior_c0 + pd.nxt, hma[i])) { dv_set_error(ctx, c->err); return -1; }
            max_mlm = std::max(max_mlm, hma[i].nlm); any_imu |= hma[i].nimu > 0; max_D = std::max(max_D, hma[i].D);
            max_fin = std::max(max_fin, be_marg_finish_smem(hma[i].D, hma[i].D - hma[i].m)); ++n_marg;
        }
    }
    const BeEvalArgs* dea = (const BeEvalArgs*)B->tab.p; const BeSolveArgs* dsa = (const BeSolveArgs*)(dea + cap);
    const BeGaugeArgs* dga = (const BeGaugeArgs*)(dsa + cap); const BeRejectArgs* drj = (const BeRejectArgs*)(dga + cap); const BeMargArgs* dma = (const BeMargArgs*)(drj + cap);
    DV_CHECK(dv_copy_async(B->tab.p, B->tab_pinned, B->tab_bytes, s));
    if (B->timing && B->tev_pending && hipEventQuery(B->tev[3]) == hipSuccess) {      // the previous round's three stages
        float ms;
        for (int k = 0; k < 3; ++k) if (hipEventElapsedTime(&ms, B->tev[k], B->tev[k + 1]) == hipSuccess) B->t_ms[k] += ms;
        B->t_n++; B->tev_pending = false;
    }
    const bool time_round = B->timing && !B->tev_pending && slots >= 3;
    // bisecting switches (dv_debug_set on the group's FIRST member; the open multi-sequence defect of round 4): one stage of the round goes through the members' own
    // single-window launches instead of the shared launch — same stream, same order, only the kernel form differs
    const int dbg = M[0]->be.debug_batch_single;      // bit 0: evaluation, 1: reduce, 2: solve
    auto be_launch_eval_batch = [&](const BeEvalArgs* t, int n, int grid, int mode, hipStream_t st) {
        if (dbg & 1) { for (dv_ctx* c : M) ::be_launch_eval(c->be.pend->ea, mode, st); } else ::be_launch_eval_batch(t, n, grid, mode, st);
    };
    auto be_launch_reduce_batch = [&](const BeSolveArgs* t, int n, int mx, int spec, hipStream_t st) {
        if (dbg & 2) { for (dv_ctx* c : M) ::be_launch_reduce(c->be.pend->sa, spec, st); } else ::be_launch_reduce_batch(t, n, mx, spec, st);
    };
    auto be_launch_solve_batch = [&](const BeSolveArgs* t, int n, int mx, int spec, hipStream_t st) -> int {
        if (dbg & 4) { for (dv_ctx* c : M) if (::be_launch_solve(c->be.pend->sa, spec, st)) return -1; return 0; }
        return ::be_launch_solve_batch(t, n, mx, spec, st);
    };
    for (int it = 0; it < slots; ++it) {                  // be_enqueue_slots' speculative schedule, one launch per stage for all windows
        const bool head = it == 0, last = it == slots - 1;
        const bool timed = time_round && it == 1 && !last;      // slot 1: solve (decision + factorisation), then the candidate's evaluation and reduce
        auto dbg_all = [&](int kind) { for (dv_ctx* c : M) be_dbg_stage(c, it, kind, s); };
        if (head) { be_launch_eval_batch(dea, S, max_grid, BE_EVAL_X, s); dbg_all(0); be_launch_reduce_batch(dsa, S, max_n, 0, s); dbg_all(1); }
        if (timed) { (void)hipEventRecord(B->tev[0], s); B->t_windows = S; }
        if (B->solve_stream) {          // (experiment: the solve on the reserved CUs — two cross-stream edges per slot)
            hipEvent_t e0 = B->ev_x[(2 * it) % B->ev_x.size()], e1 = B->ev_x[(2 * it + 1) % B->ev_x.size()];
            DV_CHECK(hipEventRecord(e0, s)); DV_CHECK(hipStreamWaitEvent(B->solve_stream, e0, 0));
            if (be_launch_solve_batch(dsa, S, max_n, head ? 0 : 1, B->solve_stream)) DV_FAIL("dv_batch_enqueue: cannot set dynamic LDS size");
            DV_CHECK(hipEventRecord(e1, B->solve_stream)); DV_CHECK(hipStreamWaitEvent(s, e1, 0));
        } else
        if (be_launch_solve_batch(dsa, S, max_n, head ? 0 : 1, s)) DV_FAIL("dv_batch_enqueue: cannot set dynamic LDS size");
        dbg_all(2);
        if (last) { be_launch_eval_batch(dea, S, max_grid, BE_EVAL_CAND_COST, s); dbg_all(3); }      // (its accept decision rides in the tail's first launch)
        else if (timed) {
            (void)hipEventRecord(B->tev[1], s);
            be_launch_eval_batch(dea, S, max_grid, BE_EVAL_CAND_FULL, s); (void)hipEventRecord(B->tev[2], s);
            be_launch_reduce_batch(dsa, S, max_n, 1, s); (void)hipEventRecord(B->tev[3], s);
            B->tev_pending = true;
        }
        else { be_launch_eval_batch(dea, S, max_grid, BE_EVAL_CAND_FULL, s); dbg_all(3); be_launch_reduce_batch(dsa, S, max_n, 1, s); dbg_all(4); }
    }
    // ---- the tails of all members: 2 + 3 launches per group instead of 5 - 6 per member on S streams ----
    // Round 4: a member's result intermittently left the single-sequence result in this launch when a second group was in flight (located by per-launch hashes,
    // scripts/dbg/multiseq_first_diff.py).  Cause: be_accept_body let thread 0 store into the control block before every wave had loaded it (be_kernels.h; fixed by a workgroup
    // barrier — shared launch 8 of 30 runs differing before, 0 of 60 after).  dv_debug_set "batch_single_tail" issues the members' own launches of the same bodies instead (A/B).
    if (dbg & 8) { for (dv_ctx* c : M) { BeGaugeArgs ga{}; be_gauge_args(c, *c->be.pend, ga); be_launch_accept_gauge(c->be.pend->sa, ga, s); } }
    else be_launch_accept_gauge_batch(dsa, dga, S, s);
    be_launch_reject_batch(drj, S, max_rej, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipEventRecord(B->ev_state, s));
    if (n_marg > 0) {
        const int rc = be_launch_marg_batch(dma, S, max_mlm, any_imu, max_D, max_fin, s);
        if (rc == -2) DV_FAIL("dv_marginalize: system does not fit in LDS");
        if (rc) DV_FAIL("dv_marginalize: cannot set dynamic LDS size");
        DV_CHECK(hipGetLastError());
    }
    for (dv_ctx* c : M) {
        BePending& pd = *c->be.pend;
        pd.deferred = false; pd.t_enq = std::chrono::steady_clock::now();
        pd.ev_state_ext = B->ev_state;
        if (pd.do_marg && !pd.pl.empty) pd.marg_in_flight = true;
    }
    B->batched_rounds++;
    return 0;
}

int be_solve_fused(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary, BeFused* fused) {
    if (be_solve_fused_begin(ctx, P, fused)) return -1;
    return be_solve_fused_end(ctx, P, summary, fused);
}

extern "C" {

// diagnostics (dv_debug_set "hash_log"): rows of seven uint64 per fused window solve: [counter, uploaded block on the device, prior A, prior b, x after the round, candidate buffer, control block]
int dv_ba_debug_dev_log(dv_ctx* ctx, unsigned long long* rows7, int cap, int* n_rows) {
    if (!ctx) return -1;
    const int n = (int)(ctx->be.dbg_dev_log.size() / 7);
    if (n_rows) *n_rows = n;
    if (rows7) std::memcpy(rows7, ctx->be.dbg_dev_log.data(), sizeof(unsigned long long) * 7 * (size_t)std::min(n, std::max(cap, 0)));
    return 0;
}

// the same per launch of the round: per fused solve DBG_SLOTS x 5 launch kinds x DBG_RANGES hashes (0 = launch not issued); *row_len = values per solve
int dv_ba_debug_slot_log(dv_ctx* ctx, unsigned long long* vals, long long cap_vals, long long* n_vals, int* row_len) {
    if (!ctx) return -1;
    const long long n = (long long)ctx->be.dbg_slot_log.size();
    if (n_vals) *n_vals = n;
    if (row_len) *row_len = BeWork::DBG_SLOTS * 5 * BeWork::DBG_RANGES;
    if (vals) std::memcpy(vals, ctx->be.dbg_slot_log.data(), sizeof(unsigned long long) * (size_t)std::min(n, std::max(cap_vals, 0ll)));
    return 0;
}

int dv_ba_solve(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary) {
    if (!ctx) return -1;
    // an operator-level solve with a host prior would overwrite the prior buffer the estimator's next frame reads from HBM
    if (ctx->est && ctx->be.prior_resident && P && P->prior && P->prior->valid && P->prior_A != ctx->be.priorA_buf[ctx->be.prior_cur])
        DV_FAIL("dv_ba_solve: this ctx's estimator holds a device-resident prior; use a separate ctx for operator-level calls");
    return be_solve_fused(ctx, P, summary, nullptr);
}

dv_batch* dv_batch_create(dv_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1 || n > 256) { dv_set_error(nullptr, "dv_batch_create: bad arguments"); return nullptr; }
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->batch || ctxs[i]->cfg.device != ctxs[0]->cfg.device || ctxs[i]->be.pend->active) { dv_set_error(nullptr, "dv_batch_create: members must be idle contexts of one device that belong to no other batch"); return nullptr; }
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) { dv_set_error(nullptr, "dv_batch_create: duplicate member"); return nullptr; }
    }
    if (hipSetDevice(ctxs[0]->cfg.device) != hipSuccess) { dv_set_error(nullptr, "dv_batch_create: hipSetDevice failed"); return nullptr; }
    dv_batch* B = new dv_batch();
    B->members.assign(ctxs, ctxs + n);
    { static std::atomic<int> next_index{0}; B->index = next_index.fetch_add(1); }
    bool ok = dv_group_stream_create(&B->stream, B->index) == hipSuccess && hipEventCreateWithFlags(&B->ev_slots, hipEventDisableTiming) == hipSuccess
              && hipEventCreateWithFlags(&B->ev_state, hipEventDisableTiming) == hipSuccess;
    if (ok && dv_solve_cus()) {
        ok = dv_solve_stream_create(&B->solve_stream) == hipSuccess;
        for (int i = 0; ok && i < 32; ++i) { hipEvent_t e = nullptr; ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; if (ok) B->ev_x.push_back(e); }
    }
    const size_t bytes = (size_t)n * (sizeof(BeEvalArgs) + sizeof(BeSolveArgs) + sizeof(BeGaugeArgs) + sizeof(BeRejectArgs) + sizeof(BeMargArgs));
    B->tab_bytes = bytes;
    ok = ok && B->tab.ensure(bytes) == hipSuccess && hipHostMalloc(&B->tab_pinned, bytes, hipHostMallocDefault) == hipSuccess;
    if (!ok) { dv_set_error(nullptr, "dv_batch_create: out of resources"); B->members.clear(); dv_batch_destroy(B); return nullptr; }
    // from now on the batch's stream IS every member's BA stream: uploads, window solves (shared or alone), tails and marginalizations of all members are ordered on
    // it — one hardware queue per group instead of one per member (48 streams on 12 queues made unrelated launches wait behind each other's event waits)
    for (int i = 0; i < n; ++i) {
        dv_ctx* c = ctxs[i];
        (void)hipStreamSynchronize(c->be_stream);
        c->be_stream_own = c->be_stream; c->be_stream = B->stream; c->batch = B;
    }
    return B;
}
void dv_batch_destroy(dv_batch* B) {
    if (!B) return;
    if (B->stream) (void)hipStreamSynchronize(B->stream);
    { std::lock_guard<std::mutex> lk(B->mu); for (dv_ctx* c : B->members) if (c->batch == B) { c->batch = nullptr; if (c->be_stream_own) { c->be_stream = c->be_stream_own; c->be_stream_own = nullptr; } } B->members.clear(); }
    if (B->front) { dv_front_batch_release(B->front); B->front = nullptr; }
    if (B->solve_stream) { (void)hipStreamSynchronize(B->solve_stream); (void)hipStreamDestroy(B->solve_stream); }
    for (hipEvent_t e : B->ev_x) (void)hipEventDestroy(e);
    if (B->stream) (void)hipStreamDestroy(B->stream);
    if (B->ev_slots) (void)hipEventDestroy(B->ev_slots);
    if (B->ev_state) (void)hipEventDestroy(B->ev_state);
    for (hipEvent_t e : B->tev) if (e) (void)hipEventDestroy(e);
    B->tab.release();
    if (B->tab_pinned) (void)hipHostFree(B->tab_pinned);
    delete B;
}
int dv_batch_enqueue(dv_batch* B) {
    if (!B) return -1;
    return batch_enqueue_impl(B);
}
// Rendezvous form for one host thread per member: every thread calls it after its member's dv_est_process_begin; the call returns in all of them once the
// last one has arrived and enqueued the round (a barrier inside the library: no interpreter lock is held while waiting).
int dv_batch_arrive(dv_batch* B) {
    if (!B) return -1;
    std::unique_lock<std::mutex> lk(B->mu);
    if (B->aborted) { dv_set_error(nullptr, "dv_batch_arrive: the batch was aborted"); return -1; }
    const long long gen = B->generation;
    if (++B->arrived >= (int)B->members.size()) {
        B->last_rc = batch_enqueue_impl(B);
        B->arrived = 0; ++B->generation;
        lk.unlock();
        B->cv.notify_all();
        return B->last_rc;
    }
    B->cv.wait(lk, [&] { return B->generation != gen; });
    return B->last_rc;
}
// A member thread that fails before it can arrive calls this (except / finally of the worker): the round is abandoned, every thread waiting in
// dv_batch_arrive — and every later arrival — returns -1 instead of blocking for ever.
int dv_batch_abort(dv_batch* B) {
    if (!B) return -1;
    {
        std::lock_guard<std::mutex> lk(B->mu);
        B->aborted = true; B->last_rc = -1; B->arrived = 0; ++B->generation;
    }
    dv_set_error(nullptr, "dv_batch_arrive: the batch was aborted");
    B->cv.notify_all();
    return 0;
}
int dv_est_get_marg_health(dv_ctx* ctx, long long* checked, long long* clamped, double* last4) {
    if (!ctx) return -1;
    if (checked) *checked = ctx->be.marg_checked;
    if (clamped) *clamped = ctx->be.marg_clamped;
    if (last4) std::memcpy(last4, ctx->be.marg_last, 32);
    return 0;
}
// per-stage launch durations of the batched window solve, HIP events on the batch stream: out3 = average ms of [be_solve_batch, be_eval_batch (full), be_reduce_batch] over
// the rounds timed so far (one steady-state slot per round), *windows = windows per launch of the last timed round.  on != 0 switches the events on.
int dv_batch_timing(dv_batch* B, int on, double* out3, long long* rounds, int* windows) {
    if (!B) return -1;
    if (on && !B->tev[0]) for (auto& e : B->tev) if (hipEventCreate(&e) != hipSuccess) return -1;
    B->timing = on != 0;
    if (out3) for (int k = 0; k < 3; ++k) out3[k] = B->t_n ? B->t_ms[k] / (double)B->t_n : 0.0;
    if (rounds) *rounds = B->t_n;
    if (windows) *windows = B->t_windows;
    return 0;
}
int dv_batch_info(dv_batch* B, long long* batched_rounds, long long* single_rounds) {
    if (!B) return -1;
    if (batched_rounds) *batched_rounds = B->batched_rounds;
    if (single_rounds) *single_rounds = B->single_rounds;
    return 0;
}

// debug-only switches (not read from the environment): "short_first_pass" = enqueue max_iters - 2 slots first so that the spare-slot
// continuation of be_solve_fused_end runs on every frame (tests/test_estimator_parity.py::test_spare_slot_path_is_equivalent)
int dv_debug_set(dv_ctx* ctx, const char* key, int value) {
    if (!ctx || !key) return -1;
    if (std::strcmp(key, "short_first_pass") == 0) { ctx->be.debug_short_first_pass = value != 0; return 0; }
    if (std::strcmp(key, "peer_timeout_ms") == 0) { ctx->dist.peer_timeout_ticks = 100000ll * std::max(value, 1); return 0; }      // transport peer: how long a wait kernel spins for a peer's flag (default 2000)
    if (std::strcmp(key, "batch_single_eval") == 0) { ctx->be.debug_batch_single = (ctx->be.debug_batch_single & ~1) | (value ? 1 : 0); return 0; }
    if (std::strcmp(key, "batch_single_reduce") == 0) { ctx->be.debug_batch_single = (ctx->be.debug_batch_single & ~2) | (value ? 2 : 0); return 0; }
    if (std::strcmp(key, "batch_single_tail") == 0) { ctx->be.debug_batch_single = (ctx->be.debug_batch_single & ~8) | (value ? 8 : 0); return 0; }
    if (std::strcmp(key, "batch_single_solve") == 0) { ctx->be.debug_batch_single = (ctx->be.debug_batch_single & ~4) | (value ? 4 : 0); return 0; }
    if (std::strcmp(key, "hash_light") == 0) { ctx->be.debug_hash_light = value != 0; return 0; }
    if (std::strcmp(key, "hash_log") == 0) { ctx->be.debug_hash_log = value != 0; return 0; }      // the estimator keeps per-solve hashes of what it uploads / downloads (dv_est_debug_hash_log)
    if (std::strcmp(key, "wait_tail") == 0) { ctx->be.debug_wait_tail = value != 0; return 0; }
    if (std::strcmp(key, "gpu_reject") == 0) { ctx->be.gpu_reject = value != 0; return 0; }      // 0: OutliersRejection on the host (rounds 1-3 until be_reject_kernel)
    if (std::strcmp(key, "c0_side") == 0) { ctx->be.c0_side = value != 0; return 0; }      // 0: the prior's constant c0 is computed on the BA stream, inside be_marg_finish (rounds 1-2)
    if (std::strcmp(key, "ldl_generic") == 0) { ctx->be.ldl_generic = value != 0; return 0; }      // the generic 4-wide panel LDL^T instead of the 16-wide MFMA form (A/B runs, agreement tests)
    DV_FAIL(std::string("dv_debug_set: unknown key ") + key);
}

int dv_ba_eval(dv_ctx* ctx, const dv_ba_problem* P, int* n_out, double* cost, double* S, double* g) {
    if (!ctx) return -1;
    if (!P || !n_out) DV_FAIL("dv_ba_eval: null argument");
    if (ctx->be.pend->active) DV_FAIL("dv_ba_eval: a solve is in flight");
    if (ctx->est && ctx->be.prior_resident && P->prior && P->prior->valid && P->prior_A != ctx->be.priorA_buf[ctx->be.prior_cur])
        DV_FAIL("dv_ba_eval: this ctx's estimator holds a device-resident prior; use a separate ctx for operator-level calls");
    if (be_begin_impl(ctx, const_cast<dv_ba_problem*>(P), nullptr, true)) return -1;
    BeWork& w = ctx->be;
    hipStream_t s = ctx->be_stream;
    const BePending& pd = *w.pend;
    if (pd.trivial) { w.pend->active = false; *n_out = 0; if (cost) *cost = 0; return 0; }
    const int n = pd.sa.dims.nstate, NBR = (n + 3) / 4, nblk = NBR * (NBR + 1) / 2;
    *n_out = n;
    std::vector<double> blk((size_t)nblk * 16), gv(2 * (size_t)n), lcost((size_t)std::max(P->nlm, 1)), io((size_t)std::max(P->nimu, 1) * IMU_OUT_STRIDE), pc(1);
    DV_CHECK(hipMemcpyAsync(blk.data(), w.Sc[0], 8 * blk.size(), hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(gv.data(), w.gvec[0], 8 * gv.size(), hipMemcpyDeviceToHost, s));
    if (P->nlm) DV_CHECK(hipMemcpyAsync(lcost.data(), w.packets[0] + (size_t)BE_PK_COST * BE_PK_STRIDE, 8 * (size_t)P->nlm, hipMemcpyDeviceToHost, s));
    if (P->nimu) DV_CHECK(hipMemcpyAsync(io.data(), w.imu_out[0], 8 * (size_t)P->nimu * IMU_OUT_STRIDE, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(pc.data(), w.prior_out[0], 8, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (be_dist_check(ctx)) return -1;
    if (cost) {
        double c = 0;
        if (pd.sa.sh.on) {          // sharded window: the landmark costs of the other ranks are known as their rank-ordered sum only (form scalar 10 of set 0)
            double lc = 0; DV_CHECK(hipMemcpy(&lc, pd.sa.sh.qf[0] + BE_QF_A + 10, 8, hipMemcpyDeviceToHost)); c = lc;
        } else for (int l = 0; l < P->nlm; ++l) c += lcost[l];
        for (int k = 0; k < P->nimu; ++k) c += io[(size_t)k * IMU_OUT_STRIDE];
        c += pc[0]; *cost = c;
    }
    if (g) for (int i = 0; i < n; ++i) g[i] = gv[i] - gv[n + i];
    if (S) for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {          // unpack the block-packed lower triangle (be_solve.hip: blk_pos)
        const int bi = i >> 2, bj = j >> 2;
        const double v = blk[((size_t)bj * NBR - (size_t)bj * (bj - 1) / 2 + bi - bj) * 16 + (i & 3) * 4 + (j & 3)];
        S[(size_t)i * n + j] = v; S[(size_t)j * n + i] = v;
    }
    return 0;
}

int dv_marginalize(dv_ctx* ctx, const dv_ba_problem* P, int mode, dv_ba_prior* out_prior, double* out_A, double* out_b, double* diag4) {
    if (!ctx) return -1;
    if (!P || !out_prior || !out_A || !out_b) DV_FAIL("dv_marginalize: null argument");
    if (ctx->be.pend->active) DV_FAIL("dv_marginalize: a solve is in flight on this ctx");
    if (ctx->est && ctx->be.prior_resident) DV_FAIL("dv_marginalize: this ctx's estimator holds a device-resident prior; use a separate ctx for operator-level calls");
    if (mode != 0 && mode != 1) DV_FAIL("dv_marginalize: mode must be 0 (kMarginOld) or 1 (kMarginSecondNew)");
    if (P->nframes != BE_NF) DV_FAIL("dv_marginalize: needs a full window (frame == kWinSize)");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (be_ensure(ctx, P->nfac)) return -1;
    BeWork& w = ctx->be;
    hipStream_t s = ctx->be_stream;
    const bool has_prior = P->prior && P->prior->valid;
    std::memset(out_prior, 0, sizeof(*out_prior));
    const int nimu = (mode == 0) ? P->nimu : 0, nlm = (mode == 0) ? P->nlm : 0, nfac = (mode == 0) ? P->nfac : 0;
    if (nimu > 1) DV_FAIL("dv_marginalize: at most the IMU factor (0,1)");
    for (int f = 0; f < nfac; ++f) if (P->factors[f].fi != 0) DV_FAIL("dv_marginalize: only landmarks anchored in frame 0 take part (estimator.cpp:446)");
    static thread_local MargPlan pl;
    std::vector<int> sel(nlm);
    for (int l = 0; l < nlm; ++l) sel[l] = l;
    if (marg_plan(ctx, pl, mode, has_prior ? P->prior : nullptr, P->factors, P->landmarks, sel.data(), nlm, nimu == 1)) return -1;
    if (pl.empty) { out_prior->valid = 0; if (diag4) diag4[0] = diag4[1] = diag4[2] = diag4[3] = 0; return 0; }
    const int n = pl.n;
    // ---- upload (pinned mirror of the upload region, one copy) ----
    uint8_t* hp = (uint8_t*)w.pinned;
    BeState* hx = (BeState*)(hp + w.up_x);
    std::memset(hx, 0, offsetof(BeState, inv_depth));
    for (int f = 0; f < BE_NF; ++f) { std::memcpy(hx->pose[f], P->pose + 7 * f, 56); if (P->use_imu) std::memcpy(hx->sb[f], P->speed_bias + 9 * f, 72); }
    std::memcpy(hx->ex, P->ex_pose, 14 * 8); hx->td = P->td[0];
    int max_lm = 0;
    for (int f = 0; f < nfac; ++f) max_lm = std::max(max_lm, P->factors[f].lm + 1);
    if (max_lm > BE_MAX_LM) DV_FAIL("dv_marginalize: landmark index out of range");
    if (max_lm) std::memcpy(hx->inv_depth, P->inv_depth, 8 * (size_t)max_lm);
    BeImu* himu = (BeImu*)(hp + w.up_imu);
    if (nimu == 1 && be_fill_imu(P->imu[0], himu[0], nullptr)) DV_FAIL("dv_marginalize: IMU covariance is singular");
    BePriorHdr ph{};
    if (has_prior) std::memcpy(&ph, P->prior, sizeof(ph));
    std::memcpy(hp + w.up_prior, &ph, sizeof(ph));
    std::memcpy(hp + w.up_mt, pl.tab, sizeof(pl.tab));
    if (nlm) std::memcpy(hp + w.up_lm, P->landmarks, sizeof(BeLm) * (size_t)nlm);
    if (nfac) std::memcpy(hp + w.up_fac, P->factors, sizeof(BeFactor) * (size_t)nfac);
    DV_CHECK(hipMemcpyAsync(w.block.p, hp, w.up_fac + sizeof(BeFactor) * (size_t)nfac, hipMemcpyHostToDevice, s));
    if (has_prior) {
        DV_CHECK(hipMemcpyAsync(w.priorA_buf[w.prior_cur], P->prior_A, 8 * (size_t)ph.n * ph.n, hipMemcpyHostToDevice, s));
        DV_CHECK(hipMemcpyAsync(w.priorb_buf[w.prior_cur], P->prior_b, 8 * (size_t)ph.n, hipMemcpyHostToDevice, s));
        w.prior_resident = false;
    }
    double* d_outA = w.Sc[0]; double* d_outb = w.gvec[0]; double* d_scal = w.marg_scal;      // Sc / gvec are idle outside a solve
    if (marg_enqueue(ctx, pl, w.x, P->g_norm, w.priorA_buf[w.prior_cur], w.priorb_buf[w.prior_cur], d_outA, d_outb, d_scal, nullptr, s)) return -1;
    DV_CHECK(hipGetLastError());
    double scal[4];
    DV_CHECK(hipMemcpyAsync(out_A, d_outA, 8 * (size_t)n * n, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(out_b, d_outb, 8 * (size_t)n, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(scal, d_scal, 32, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (ctx->timing) dv_harvest_timers(ctx, s);
    if (diag4) std::memcpy(diag4, scal, 32);
    std::memcpy(w.marg_last, scal, 32); w.marg_checked++;
    if (scal[2] != 0.0) w.marg_clamped++;          // pivots <= 1e-8 skipped on the device (pseudo-inverse like the reference's eigen clamp); reported through diag4[2]
    marg_new_prior(pl, P->pose, P->speed_bias, P->ex_pose, P->td, scal[0], out_prior);
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
    if (be_fill_imu(*imu, h, nullptr)) DV_FAIL("dv_imu_eval: IMU covariance is singular");
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
