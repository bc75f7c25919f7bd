// dvins_api.hip — C ABI (include/dvins.h) of libdvins_hip.so: context, HBM layout and per-frame
// orchestration of the front end.  Everything a frame needs lives in HBM inside the ctx:
//   * three image pyramids (left current, left previous, right), each level pitched to 16 B,
//     level 0 is a pitched copy of the input written by the first pyrDown launch;
//   * the tracker state as a struct of arrays (DvTrackState) with its counters in device memory,
//     so the whole of FeatureTracker::TrackImage is enqueued without a host round trip;
//   * Shi-Tomasi candidate records + the pinned host staging buffer for the frame's output.
// One frame = pyrDown x3 (stereo pair per launch) -> LK temporal -> compact/sort -> Shi-Tomasi tile
// -> select/append -> LK stereo -> finalize -> one D2H copy of <= max_cnt 128-byte rows.
#include "dv_ctx.h"

static std::string g_last_error;
static std::mutex g_err_mutex;

void dv_set_error(dv_ctx* ctx, const std::string& msg) {
    if (ctx) { std::lock_guard<std::mutex> lk(ctx->err_mu); ctx->err = msg; }
    std::lock_guard<std::mutex> lk(g_err_mutex);
    g_last_error = msg;
}

// disc half-widths of cv::circle's midpoint rasteriser (drawing.cpp Circle(), fill = true)
static void circle_half_widths(int radius, std::vector<uint8_t>& hw) {
    hw.assign(radius + 1, 0);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hw[dy] = (uint8_t)std::max<int>(hw[dy], dx);
        hw[dx] = (uint8_t)std::max<int>(hw[dx], dy);
        dy++; err += plus; plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
}

static int ensure_hw(dv_ctx* ctx, int radius) {
    if (radius < 0 || radius > DV_MAX_RADIUS) DV_FAIL("disc radius (min_dist) out of range [0,128]");
    if (ctx->hw_radius == radius) return 0;
    std::vector<uint8_t> hw; circle_half_widths(radius, hw);
    DV_CHECK(ctx->hw_buf.ensure(DV_MAX_RADIUS + 1));
    DV_CHECK(hipMemcpyAsync(ctx->hw_buf.p, hw.data(), hw.size(), hipMemcpyHostToDevice, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->hw_radius = radius;
    return 0;
}

static int ensure_cand(dv_ctx* ctx, int w, int h) {
    int cap = std::max(4096, (w * h) / 4);
    if (cap <= ctx->cand_cap) return 0;
    DV_CHECK(ctx->cand_buf.ensure((size_t)cap * sizeof(DvCand)));
    ctx->cand_cap = cap;
    return 0;
}

StageTimer* dv_timer_for(dv_ctx* ctx, const char* name) {
    std::lock_guard<std::mutex> lk(ctx->timer_mu);
    for (auto& t : ctx->timers) if (t.name == name) return &t;
    ctx->timers.emplace_back();
    StageTimer& t = ctx->timers.back();
    t.name = name;
    return &t;
}
void dv_harvest_timers(dv_ctx* ctx, hipStream_t synced) {
    std::lock_guard<std::mutex> lk(ctx->timer_mu);
    for (auto& t : ctx->timers) {
        if (t.stream != synced) continue;
        for (size_t i = 0; i < t.used; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.pool[i].first, t.pool[i].second) == hipSuccess) { t.total_ms += ms; t.count++; }
        }
        t.used = 0;
    }
}

// Builds the pyramids of one image (img1 == nullptr) or of a stereo pair with shared launches.
static int build_pyramids(dv_ctx* ctx, PyrSet& P0, PyrSet* P1, const uint8_t* img0, const uint8_t* img1, int w, int h, int stride,
                          int mem, int max_level, bool undistort = false) {
    DV_CHECK(P0.alloc(w, h, max_level));
    if (P1) DV_CHECK(P1->alloc(w, h, max_level));
    const DvPyr& a = P0.pyr;
    const DvPyr* b = P1 ? &P1->pyr : nullptr;
    hipStream_t s = ctx->stream;
    const bool bgr = (mem & DV_FMT_BGR) != 0;
    mem &= ~DV_FMT_BGR;
    const bool dev = (mem == DV_MEM_DEVICE);
    if (undistort) {    // cfg::is_undistort_input: cv::remap (+ cvtColor for colour frames) straight into level 0; host frames are staged in HBM first
        if (w != ctx->undist_w || h != ctx->undist_h) DV_FAIL("dv_track_stereo: undistortion maps were installed for another image size");
        if (b && !ctx->undist[1]) DV_FAIL("dv_track_stereo: undistortion maps installed for camera 0 but not for camera 1");
        const int cn = bgr ? 3 : 1;
        const uint8_t* c0 = img0; const uint8_t* c1 = img1; int cp = stride;
        if (!dev) {
            cp = align_up(cn * w, 16);
            DV_CHECK(ctx->s3.ensure((size_t)cp * h)); DV_CHECK(hipMemcpy2DAsync(ctx->s3.p, cp, img0, stride, (size_t)cn * w, h, hipMemcpyHostToDevice, s));
            c0 = (const uint8_t*)ctx->s3.p;
            if (b) { DV_CHECK(ctx->s4.ensure((size_t)cp * h)); DV_CHECK(hipMemcpy2DAsync(ctx->s4.p, cp, img1, stride, (size_t)cn * w, h, hipMemcpyHostToDevice, s)); c1 = (const uint8_t*)ctx->s4.p; }
        }
        const size_t m2off = (size_t)4 * w * h;
        const uint8_t* mb0 = (const uint8_t*)ctx->undist_buf[0].p; const uint8_t* mb1 = (const uint8_t*)ctx->undist_buf[1].p;
        dv_launch_remap(c0, b ? c1 : nullptr, w, h, cp, cn, bgr ? 1 : 0, (const int16_t*)mb0, (const uint16_t*)(mb0 + m2off),
                        b ? (const int16_t*)mb1 : nullptr, b ? (const uint16_t*)(mb1 + m2off) : nullptr, a.L[0].p, b ? b->L[0].p : nullptr, a.L[0].pitch, s);
    } else if (bgr) {   // colour input: BGR -> gray straight into level 0 (row N2); host frames are staged in HBM first
        const uint8_t* c0 = img0; const uint8_t* c1 = img1; int cp = stride;
        if (!dev) {
            cp = align_up(3 * w, 16);
            DV_CHECK(ctx->s3.ensure((size_t)cp * h)); DV_CHECK(hipMemcpy2DAsync(ctx->s3.p, cp, img0, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, s));
            c0 = (const uint8_t*)ctx->s3.p;
            if (b) { DV_CHECK(ctx->s4.ensure((size_t)cp * h)); DV_CHECK(hipMemcpy2DAsync(ctx->s4.p, cp, img1, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, s)); c1 = (const uint8_t*)ctx->s4.p; }
        }
        dv_launch_bgr2gray(c0, b ? c1 : nullptr, w, h, cp, a.L[0].p, b ? b->L[0].p : nullptr, a.L[0].pitch, s);
    } else if (!dev || a.levels == 1) {
        hipMemcpyKind k = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        DV_CHECK(hipMemcpy2DAsync(a.L[0].p, a.L[0].pitch, img0, stride, w, h, k, s));
        if (b) DV_CHECK(hipMemcpy2DAsync(b->L[0].p, b->L[0].pitch, img1, stride, w, h, k, s));
    }
    for (int l = 1; l < a.levels; ++l) {
        const bool fuse_copy = dev && l == 1 && !bgr && !undistort;
        const uint8_t* s0 = fuse_copy ? img0 : a.L[l - 1].p;
        const uint8_t* s1 = b ? (fuse_copy ? img1 : b->L[l - 1].p) : nullptr;
        const int sp = fuse_copy ? stride : a.L[l - 1].pitch;
        dv_launch_pyr_down2(s0, s1, a.L[l - 1].w, a.L[l - 1].h, sp, a.L[l].p, b ? b->L[l].p : nullptr, a.L[l].pitch,
                            fuse_copy ? a.L[0].p : nullptr, (fuse_copy && b) ? b->L[0].p : nullptr, a.L[0].pitch, s);
    }
    dv_launch_pyr_apron(a, b, s);
    DV_CHECK(hipGetLastError());
    return 0;
}

// the pyramid cv::cuda::SparsePyrLKOpticalFlow builds (cuda::pyrDown: round half to even) on top of an existing level 0 (a0 / b0: level 0 of the regular pyramids)
static int build_cuda_pyramids(dv_ctx* ctx, PyrSet& C0, PyrSet* C1, const DvPyr& a0, const DvPyr* b0, int w, int h, int max_level) {
    DV_CHECK(C0.alloc(w, h, max_level, true));
    if (C1) DV_CHECK(C1->alloc(w, h, max_level, true));
    C0.pyr.L[0] = a0.L[0];
    if (C1) C1->pyr.L[0] = b0->L[0];
    for (int l = 1; l < C0.pyr.levels; ++l)
        dv_launch_pyr_down2(C0.pyr.L[l - 1].p, C1 ? C1->pyr.L[l - 1].p : nullptr, C0.pyr.L[l - 1].w, C0.pyr.L[l - 1].h, C0.pyr.L[l - 1].pitch, C0.pyr.L[l].p, C1 ? C1->pyr.L[l].p : nullptr,
                            C0.pyr.L[l].pitch, nullptr, nullptr, 0, ctx->stream, 1);
    DV_CHECK(hipGetLastError());
    return 0;
}

extern "C" {

void* dv_pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { dv_set_error(nullptr, "dv_pinned_alloc: hipHostMalloc failed"); return nullptr; }
    return p;
}
void dv_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

const char* dv_last_error(dv_ctx* ctx) {
    if (ctx) {      // a copy per calling thread: another thread of a dynamic sequence may be writing the ctx's string (valid until this thread's next call)
        static thread_local std::string mine;
        std::lock_guard<std::mutex> lk(ctx->err_mu);
        mine = ctx->err;
        return mine.c_str();
    }
    std::lock_guard<std::mutex> lk(g_err_mutex);
    return g_last_error.c_str();
}

dv_ctx* dv_create(const dv_config* cfg) {
    if (!cfg) { dv_set_error(nullptr, "dv_create: null config"); return nullptr; }
    if (cfg->width <= 0 || cfg->height <= 0) { dv_set_error(nullptr, "dv_create: bad image size"); return nullptr; }
    if (cfg->max_cnt <= 0 || cfg->max_cnt > DV_MAX_FEATS) { dv_set_error(nullptr, "dv_create: max_cnt must be in [1,1024]"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { dv_set_error(nullptr, "dv_create: no HIP device (the HIP path has no CPU fallback)"); return nullptr; }
    if (cfg->device < 0 || cfg->device >= ndev) { dv_set_error(nullptr, "dv_create: bad device ordinal"); return nullptr; }
    dv_ctx* ctx = new dv_ctx();
    ctx->cfg = *cfg;
    auto fail = [&](const char* what, hipError_t e) -> dv_ctx* {
        dv_set_error(nullptr, std::string("dv_create: ") + what + ": " + hipGetErrorString(e));
        delete ctx; return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(cfg->device)) != hipSuccess) return fail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&ctx->be_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipStreamCreateWithFlags(&ctx->obj_stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->done, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_pyr, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&ctx->ev_bg_select, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
    // tracker state: one block, struct of arrays
    const size_t N = DV_MAX_FEATS;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    size_t o_last = take(N * 8), o_cur = take(N * 8), o_lk = take(N * 8), o_lks = take(N), o_ids = take(N * 4), o_cnt = take(N * 4),
           o_pun = take(N * 8), o_prun = take(N * 8), o_prv = take(N), o_trk = take(N), o_rp = take(N * 8), o_rs = take(N),
           o_scal = take(64), o_ord = take(N * 2);
    if ((e = ctx->state_block.ensure(off)) != hipSuccess) return fail("hipMalloc(state)", e);
    uint8_t* base = (uint8_t*)ctx->state_block.p;
    ctx->tr.last_pts = (float2*)(base + o_last); ctx->tr.curr_pts = (float2*)(base + o_cur); ctx->tr.lk_pts = (float2*)(base + o_lk);
    ctx->tr.lk_status = base + o_lks; ctx->tr.ids = (uint32_t*)(base + o_ids); ctx->tr.track_cnt = (int32_t*)(base + o_cnt);
    ctx->tr.prev_un = (float2*)(base + o_pun); ctx->tr.prev_run = (float2*)(base + o_prun); ctx->tr.prev_rvalid = base + o_prv;
    ctx->tr.tracked = base + o_trk; ctx->tr.right_pts = (float2*)(base + o_rp); ctx->tr.right_status = base + o_rs;
    ctx->tr.n_feat = (int*)(base + o_scal); ctx->tr.n_tracked = ctx->tr.n_feat + 1; ctx->tr.next_id = (uint32_t*)(ctx->tr.n_feat + 2);
    ctx->tr.lk_order = (unsigned short*)(base + o_ord);
    ctx->n_cand = ctx->tr.n_feat + 3; ctx->max_ord = (unsigned*)(ctx->tr.n_feat + 4); ctx->err_flag = ctx->tr.n_feat + 5;
    if ((e = ctx->out_buf.ensure(N * sizeof(dv_feat) + 256)) != hipSuccess) return fail("hipMalloc(out)", e);
    ctx->out_dev = (dv_feat*)ctx->out_buf.p; ctx->nout_dev = (int*)((uint8_t*)ctx->out_buf.p + N * sizeof(dv_feat));
    void* pinned = nullptr;
    if ((e = hipHostMalloc(&pinned, N * sizeof(dv_feat) + 256, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
    ctx->out_pinned = (dv_feat*)pinned; ctx->nout_pinned = (int*)((uint8_t*)pinned + N * sizeof(dv_feat)); ctx->err_pinned = ctx->nout_pinned + 1;
    if (dv_reset(ctx) != 0) { std::string m = ctx->err; delete ctx; dv_set_error(nullptr, m); return nullptr; }
    return ctx;
}

void dv_destroy(dv_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    be_batch_detach(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    // (a frame tracked on a dv_batch's front-end stream: be_batch_detach above / dv_batch_destroy drain that stream — its event belongs to the batch and may be gone by now,
    //  so it is NOT waited on here: doing so crashed dv_destroy behind dv_runner_destroy, round 5)
    if (ctx->be_stream) (void)hipStreamSynchronize(ctx->be_stream);
    for (auto& t : ctx->timers) for (auto& p : t.pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (DevBuf* b : { &ctx->state_block, &ctx->cand_buf, &ctx->hw_buf, &ctx->mask_buf, &ctx->out_buf, &ctx->s0, &ctx->s1, &ctx->s2, &ctx->s3, &ctx->s4,
                       &ctx->left[0].buf, &ctx->left[1].buf, &ctx->right.buf, &ctx->leftc[0].buf, &ctx->leftc[1].buf, &ctx->rightc.buf, &ctx->opA.buf, &ctx->opB.buf, &ctx->undist_buf[0], &ctx->undist_buf[1] }) b->release();
    if (ctx->inst) dv_inst_destroy_internal(ctx->inst);
    if (ctx->est) dv_est_destroy_internal(ctx->est);
    be_dist_release(ctx);
    if (ctx->be.c0_stream) { (void)hipStreamSynchronize(ctx->be.c0_stream); (void)hipStreamDestroy(ctx->be.c0_stream); }
    if (ctx->be.ev_margA) (void)hipEventDestroy(ctx->be.ev_margA);
    if (ctx->be.ev_c0) (void)hipEventDestroy(ctx->be.ev_c0);
    ctx->be.block.release(); ctx->be.marg_buf.release();
    if (ctx->be.pinned) (void)hipHostFree(ctx->be.pinned);
    if (ctx->be.rej_pinned) (void)hipHostFree(ctx->be.rej_pinned);
    ctx->obj_buf.release(); ctx->obj_pend.release(); ctx->obj_op_pend.release();
    if (ctx->obj_stream) (void)hipStreamDestroy(ctx->obj_stream);
    if (ctx->be_stream) (void)hipStreamDestroy(ctx->be_stream);
    if (ctx->out_pinned) (void)hipHostFree(ctx->out_pinned);
    if (ctx->unmask_pinned) (void)hipHostFree(ctx->unmask_pinned);
    if (ctx->done) (void)hipEventDestroy(ctx->done);
    if (ctx->ev_pyr) (void)hipEventDestroy(ctx->ev_pyr);
    if (ctx->ev_bg_select) (void)hipEventDestroy(ctx->ev_bg_select);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dv_reset(dv_ctx* ctx) {
    if (!ctx) return -1;
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    DV_CHECK(hipMemsetAsync(ctx->state_block.p, 0, ctx->state_block.bytes, ctx->stream));
    const uint32_t one = 1;    // InstFeat::global_id_count{1} (front_end/instance_feature.h:137)
    DV_CHECK(hipMemcpyAsync(ctx->tr.next_id, &one, 4, hipMemcpyHostToDevice, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->last_front && ctx->last_front != ctx->stream && ctx->last_done) DV_CHECK(hipEventSynchronize(ctx->last_done));
    ctx->have_prev = false; ctx->prev_time = 0.0; ctx->pending = false; ctx->last_done = nullptr; ctx->last_front = nullptr; ctx->leftc_valid[0] = ctx->leftc_valid[1] = false;
    if (ctx->inst && dv_inst_reset(ctx)) return -1;
    return 0;
}

int dv_sync(dv_ctx* ctx) {
    if (!ctx) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_timing_enable(dv_ctx* ctx, int on) { if (!ctx) return -1; ctx->host_timing = on == -1; ctx->timing = on > 0; ctx->kernel_timing = on >= 2; return 0; }      // -1: host wall-clock scopes only (no events, no extra syncs)
int dv_timing_reset(dv_ctx* ctx) { if (!ctx) return -1; std::lock_guard<std::mutex> lk(ctx->timer_mu); for (auto& t : ctx->timers) { t.total_ms = 0; t.count = 0; t.used = 0; } return 0; }
int dv_timing_get(dv_ctx* ctx, const char* name, double* total_ms, long long* count) {
    if (!ctx || !name) return -1;
    std::lock_guard<std::mutex> lk(ctx->timer_mu);
    for (auto& t : ctx->timers) if (t.name == name) { if (total_ms) *total_ms = t.total_ms; if (count) *count = t.count; return 0; }
    if (total_ms) *total_ms = 0; if (count) *count = 0;
    return 0;
}

int dv_track_stereo_enqueue(dv_ctx* ctx, const uint8_t* gray0, const uint8_t* gray1, int w, int h, int stride, double t,
                            const uint8_t* mask_or_null, int mode, int mem) {
    if (!ctx) return -1;
    HostScope hs(ctx, "h_front_enqueue");
    if (!gray0) DV_FAIL("dv_track_stereo: gray0 is null");
    if (w != ctx->cfg.width || h != ctx->cfg.height) DV_FAIL("dv_track_stereo: image size differs from config (reference: std::terminate, main.cpp:95-99)");
    if (ctx->pending) DV_FAIL("dv_track_stereo_enqueue: previous frame not collected");
    if (mode != DV_MODE_RAW && mode != DV_MODE_NAIVE && mode != DV_MODE_SEMANTIC) DV_FAIL("dv_track_stereo: unknown mode");
    if ((mem & 0xff) == DV_MEM_PINNED) mem = (mem & ~0xff) | DV_MEM_DEVICE;      // pinned + mapped host memory is device-addressable: the kernels read it in place
    if ((mem & 0xff) != DV_MEM_HOST && (mem & 0xff) != DV_MEM_DEVICE) DV_FAIL("dv_track_stereo: unknown memory kind");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const dv_config& c = ctx->cfg;
    const bool stereo = c.stereo && gray1;
    hipStream_t s = ctx->stream;
    if (ensure_hw(ctx, c.min_dist)) return -1;
    if (ensure_cand(ctx, w, h)) return -1;
    if (dv_inst_wait_before_next_frame(ctx)) DV_FAIL("dv_track_stereo: hipStreamWaitEvent");
    if (ctx->last_front && ctx->last_front != s && ctx->last_done) DV_CHECK(hipStreamWaitEvent(s, ctx->last_done, 0));      // the previous frame ran on a dv_batch's front-end stream
    StageScope frame(ctx, "frame");
    ctx->cur ^= 1;
    PyrSet& L = ctx->left[ctx->cur];
    PyrSet& Lp = ctx->left[ctx->cur ^ 1];
    {
        StageScope sc(ctx, "pyr");
        if (build_pyramids(ctx, L, stereo ? &ctx->right : nullptr, gray0, stereo ? gray1 : nullptr, w, h, stride, mem, 3, ctx->undist[0])) return -1;
    }
    if (ctx->inst) DV_CHECK(hipEventRecord(ctx->ev_pyr, s));
    // FeatureTrackByLKGpu's own pyramids where the reference runs that tracker (naive: TrackLeftGPU + TrackRightGPU; semantic: TrackRightGPU)
    ctx->leftc_valid[ctx->cur] = false;
    if (mode != DV_MODE_RAW) {
        StageScope sc(ctx, "pyr_cuda");
        if (build_cuda_pyramids(ctx, ctx->leftc[ctx->cur], stereo ? &ctx->rightc : nullptr, L.pyr, stereo ? &ctx->right.pyr : nullptr, w, h, 3)) return -1;
        ctx->leftc_valid[ctx->cur] = true;
        if (mode == DV_MODE_NAIVE && ctx->have_prev && !ctx->leftc_valid[ctx->cur ^ 1]) {      // the previous frame was tracked in another mode: its pyramid of this flavour does not exist yet
            if (build_cuda_pyramids(ctx, ctx->leftc[ctx->cur ^ 1], nullptr, Lp.pyr, nullptr, w, h, 3)) return -1;
            ctx->leftc_valid[ctx->cur ^ 1] = true;
        }
    }
    const uint8_t* mask_dev = nullptr; int mask_pitch = 0;
    if (mask_or_null) {
        const bool bgr_in = (mem & DV_FMT_BGR) != 0;
        if ((mem & ~DV_FMT_BGR) == DV_MEM_DEVICE) { mask_dev = mask_or_null; mask_pitch = bgr_in ? w : stride; }
        else {
            mask_pitch = align_up(w, 16);
            DV_CHECK(ctx->mask_buf.ensure((size_t)mask_pitch * h));
            DV_CHECK(hipMemcpy2DAsync(ctx->mask_buf.p, mask_pitch, mask_or_null, bgr_in ? w : stride, w, h, hipMemcpyHostToDevice, s));
            mask_dev = (const uint8_t*)ctx->mask_buf.p;
        }
    }
    if (!ctx->unmask.empty()) {          // system/main.cpp:217-245: the static instances' pixels leave the merged mask (inv_merge_mask = 255 there) before anything reads it
        if (!mask_dev) { ctx->unmask.clear(); DV_FAIL("dv_track_unmask_static: the frame carries no mask"); }
        if (mask_dev != (const uint8_t*)ctx->mask_buf.p) {          // the caller's device buffer is not written to: work on a copy
            const int mp = align_up(w, 16);
            DV_CHECK(ctx->mask_buf.ensure((size_t)mp * h));
            DV_CHECK(hipMemcpy2DAsync(ctx->mask_buf.p, mp, mask_dev, mask_pitch, w, h, hipMemcpyDeviceToDevice, s));
            mask_dev = (const uint8_t*)ctx->mask_buf.p; mask_pitch = mp;
        }
        for (const dv_ctx::UnmaskJob& j : ctx->unmask)
            dv_launch_unmask((uint8_t*)ctx->mask_buf.p, mask_pitch, w, h, j.x, j.y, j.w, j.h, (const uint8_t*)ctx->unmask_pinned + j.off, s);
        ctx->unmask.clear();
    }
    const bool naive = (mode != DV_MODE_RAW);            // naive and semantic share the InstFeat code path (mask test, no sort, >= 10 new)
    if (naive && mask_dev && c.mask_morphology_size > 0) {       // ErodeMask (background_tracker.cpp:408-416,764-768)
        const int ep = align_up(w, 16);
        DV_CHECK(ctx->s3.ensure((size_t)ep * h)); DV_CHECK(ctx->s4.ensure((size_t)ep * h));
        dv_launch_erode(mask_dev, w, h, mask_pitch, c.mask_morphology_size, (uint8_t*)ctx->s3.p, ep, (uint8_t*)ctx->s4.p, ep, s);
        mask_dev = (const uint8_t*)ctx->s4.p; mask_pitch = ep;
    }
    // forward/backward consistency: FeatureTrackByLK keeps <= 0.5 px (feature_utils.cpp:56), FeatureTrackByLKGpu <= 1.0 px (:126) (Q12) — and the two are different
    // trackers (lk.hip / lk_cuda.hip), each used where the reference uses it
    const float dist_temporal = (mode == DV_MODE_NAIVE) ? 1.0f : 0.5f;       // TrackLeftGPU (naive) vs TrackLeft (raw, semantic)
    const float dist_stereo = (mode == DV_MODE_RAW) ? 0.5f : 1.0f;           // TrackRightGPU in naive and semantic
    if (ctx->have_prev) {
        StageScope sc(ctx, "lk_temporal");
        if (mode == DV_MODE_NAIVE)      // TrackLeftGPU -> FeatureTrackByLKGpu (instance_feature.cpp:191-216): the GPU tracker's rule
            dv_launch_lk_cuda_track(ctx->leftc[ctx->cur ^ 1].pyr, ctx->leftc[ctx->cur].pyr, ctx->tr.last_pts, ctx->tr.n_feat, c.max_cnt, c.flow_back, dist_temporal, ctx->tr.lk_pts, ctx->tr.lk_status, s);
        else
            dv_launch_lk_track(Lp.pyr, L.pyr, ctx->tr.last_pts, ctx->tr.n_feat, c.max_cnt, c.flow_back, dist_temporal, ctx->tr.lk_pts,
                               ctx->tr.lk_status, s, ctx->tr.lk_order);
    }
    {
        StageScope sc(ctx, "compact");
        dv_launch_compact(ctx->tr, naive ? mask_dev : nullptr, mask_pitch, naive ? 0 : 1, ctx->n_cand, ctx->max_ord, s);
    }
    const int min_new = naive ? 10 : 1;                   // Q23: instance_feature.cpp:353-356 vs background_tracker.cpp:82-90
    // DetectNewFeature(img, use_gpu, ...): TrackImageNaive passes true (background_tracker.cpp:445) -> DetectShiTomasiCornersGpu (feature_utils.cpp:339-348),
    // TrackSemanticImage passes false (:789) and TrackImage calls cv::goodFeaturesToTrack itself (:85)
    const int gftt_rule = (mode == DV_MODE_NAIVE) ? DV_GFTT_RULE_CUDA : DV_GFTT_RULE_CPU;
    {
        StageScope sc(ctx, "gftt_eig");
        GfttTileArgs a{};
        a.img = L.pyr.L[0].p; a.w = w; a.h = h; a.pitch = L.pyr.L[0].pitch;
        a.in_mask = mask_dev; a.mask_pitch = mask_pitch;
        a.disc_pts = ctx->tr.curr_pts; a.n_disc = ctx->tr.n_tracked; a.radius = c.min_dist; a.hw = (const uint8_t*)ctx->hw_buf.p;
        a.n_feat = ctx->tr.n_feat; a.max_cnt = c.max_cnt; a.min_new = min_new;
        a.eig_out = nullptr; a.eig_pitch = 0;
        a.cand = (DvCand*)ctx->cand_buf.p; a.cand_cap = ctx->cand_cap; a.n_cand = ctx->n_cand; a.max_ord = ctx->max_ord;
        a.rule = gftt_rule;
        dv_launch_gftt_tile(a, s);
    }
    {
        StageScope sc(ctx, "gftt_select");
        GfttSelectArgs a{};
        a.cand = (const DvCand*)ctx->cand_buf.p; a.n_cand = ctx->n_cand; a.cand_cap = ctx->cand_cap; a.max_ord = ctx->max_ord;
        a.w = w; a.h = h; a.quality = 0.01; a.min_dist = (double)c.min_dist;
        a.max_n_host = 0; a.n_feat = ctx->tr.n_feat; a.max_cnt = c.max_cnt; a.min_new = min_new;
        a.out_xy = nullptr; a.n_out = nullptr; a.tr = ctx->tr; a.has_tr = 1; a.err_flag = ctx->err_flag; a.rule = gftt_rule;
        if (dv_launch_gftt_select(a, s)) DV_FAIL("gftt_select: cannot set dynamic LDS size");
    }
    if (ctx->inst) DV_CHECK(hipEventRecord(ctx->ev_bg_select, s));
    if (stereo) {
        StageScope sc(ctx, "lk_stereo");
        if (mode != DV_MODE_RAW)        // TrackRightGPU -> FeatureTrackByLKGpu (instance_feature.cpp:278-310) in naive and semantic mode
            dv_launch_lk_cuda_track(ctx->leftc[ctx->cur].pyr, ctx->rightc.pyr, ctx->tr.curr_pts, ctx->tr.n_feat, c.max_cnt, c.flow_back, dist_stereo, ctx->tr.right_pts, ctx->tr.right_status, s);
        else
            dv_launch_lk_track(L.pyr, ctx->right.pyr, ctx->tr.curr_pts, ctx->tr.n_feat, c.max_cnt, c.flow_back, dist_stereo,
                               ctx->tr.right_pts, ctx->tr.right_status, s, ctx->tr.lk_order);
    }
    {
        StageScope sc(ctx, "finalize");
        // the rows, their count and the device error flags go straight into the pinned buffer (three copy dispatches behind the kernel before: ~20 us of the
        // tracker's latency per frame)
        dv_launch_finalize(ctx->tr, c.cam0, c.cam1, stereo ? 1 : 0, t - ctx->prev_time, c.max_cnt, ctx->out_pinned, ctx->nout_pinned, s, ctx->err_flag, ctx->err_pinned);
    }
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipEventRecord(ctx->done, s));
    ctx->last_done = ctx->done; ctx->last_front = s;
    ctx->prev_time = t; ctx->have_prev = true; ctx->pending = true;
    return 0;
}

int dv_track_unmask_static(dv_ctx* ctx, const dv_inst_det* dets, int n_dets, const uint32_t* static_ids, int n_static) {
    if (!ctx) return -1;
    ctx->unmask.clear();
    if (n_static <= 0 || n_dets <= 0) return 0;
    if (!dets || !static_ids) DV_FAIL("dv_track_unmask_static: null argument");
    if (ctx->pending) DV_FAIL("dv_track_unmask_static: call it before dv_track_stereo_enqueue of the frame it belongs to");
    size_t need = 0;
    for (int i = 0; i < n_dets; ++i) if (std::find(static_ids, static_ids + n_static, dets[i].track_id) != static_ids + n_static) need += ((size_t)dets[i].w * dets[i].h + 15) / 16 * 16;
    if (!need) return 0;
    if (ctx->unmask_pinned_bytes < need) {
        DV_CHECK(hipStreamSynchronize(ctx->stream));          // (a previous frame's kernels may still read the old staging area)
        if (ctx->unmask_pinned) (void)hipHostFree(ctx->unmask_pinned);
        ctx->unmask_pinned = nullptr; ctx->unmask_pinned_bytes = 0;
        const size_t want = std::max<size_t>(2 * need, (size_t)ctx->cfg.width * ctx->cfg.height);      // (a frame's worth from the start: the rectangles grow as objects come closer)
        DV_CHECK(hipHostMalloc(&ctx->unmask_pinned, want, hipHostMallocDefault));
        ctx->unmask_pinned_bytes = want;
    } else if (ctx->last_done && ctx->last_done == ctx->done) DV_CHECK(hipEventSynchronize(ctx->done));      // the previous frame's unmask kernels have read the staging area (a no-op wait in the usual case: that frame was collected)
    size_t off = 0;
    for (int i = 0; i < n_dets; ++i) {
        const dv_inst_det& d = dets[i];
        if (std::find(static_ids, static_ids + n_static, d.track_id) == static_ids + n_static) continue;
        if (!d.mask || d.w <= 0 || d.h <= 0 || d.x < 0 || d.y < 0 || d.x + d.w > ctx->cfg.width || d.y + d.h > ctx->cfg.height) { ctx->unmask.clear(); DV_FAIL("dv_track_unmask_static: bad detection rectangle / mask"); }
        std::memcpy((uint8_t*)ctx->unmask_pinned + off, d.mask, (size_t)d.w * d.h);
        ctx->unmask.push_back({ d.x, d.y, d.w, d.h, off });
        off += ((size_t)d.w * d.h + 15) / 16 * 16;
    }
    return 0;
}

int dv_track_stereo_collect(dv_ctx* ctx, dv_feat* out, int* n_out) {
    if (!ctx) return -1;
    if (!ctx->pending) DV_FAIL("dv_track_stereo_collect: nothing enqueued");
    { HostScope hs(ctx, "h_front_wait"); DV_CHECK(hipEventSynchronize(ctx->last_done ? ctx->last_done : ctx->done)); }
    ctx->pending = false;
    if (ctx->timing) dv_harvest_timers(ctx, ctx->stream);
    if (*ctx->err_pinned) {
        int f = *ctx->err_pinned;
        DV_CHECK(hipMemsetAsync(ctx->err_flag, 0, 4, ctx->stream));
        DV_FAIL(std::string("front end device error flags=") + std::to_string(f) +
                " (1: candidate buffer overflow, 2: min-distance grid too large, 4: value bin overflow)");
    }
    const int n = *ctx->nout_pinned;
    if (n_out) *n_out = n;
    if (out && n > 0) std::memcpy(out, ctx->out_pinned, (size_t)n * sizeof(dv_feat));
    return 0;
}

int dv_track_stereo(dv_ctx* ctx, const uint8_t* gray0, const uint8_t* gray1, int w, int h, int stride, double t,
                    const uint8_t* mask_or_null, int mode, int mem, dv_feat* out, int* n_out) {
    if (dv_track_stereo_enqueue(ctx, gray0, gray1, w, h, stride, t, mask_or_null, mode, mem)) return -1;
    return dv_track_stereo_collect(ctx, out, n_out);
}

}  // extern "C"

// ------------------------------- the front ends of a dv_batch group in shared launches -------------------------------
// FeatureTracker::TrackImage (background_tracker.cpp:52-158) of S independent sequences, one launch per STAGE for all of them (the reference runs one process per
// sequence, system/main.cpp:178-330): pyrDown levels 1..3 (level 1 also writes the pitched level-0 copy), aprons, temporal LK, compaction / sort, Shi-Tomasi tile,
// corner selection, stereo LK, rows = 10 launches per group and frame instead of 10 per sequence.  The kernels are the single-sequence kernels' bodies behind a
// job table in HBM (blockIdx.z / .y / .x = member), so every member's rows are bit-identical to what its own dv_track_stereo_enqueue produces.
struct DvFrontBatch {
    hipStream_t stream = nullptr; hipEvent_t done = nullptr, ev_copy[2] = { nullptr, nullptr };
    DevBuf tab[2]; void* tab_pinned[2] = { nullptr, nullptr }; size_t tab_bytes = 0; int parity = 0; bool copy_used[2] = { false, false };
    long long rounds = 0, members_batched = 0, members_single = 0;
};
void dv_front_batch_sync(DvFrontBatch* F) { if (F && F->stream) (void)hipStreamSynchronize(F->stream); }      // be_batch_detach: a member leaves while one of its frames may be in flight on the group's front-end stream
void dv_front_batch_release(DvFrontBatch* F) {
    if (!F) return;
    if (F->stream) { (void)hipStreamSynchronize(F->stream); (void)hipStreamDestroy(F->stream); }
    if (F->done) (void)hipEventDestroy(F->done);
    for (int k = 0; k < 2; ++k) { if (F->ev_copy[k]) (void)hipEventDestroy(F->ev_copy[k]); F->tab[k].release(); if (F->tab_pinned[k]) (void)hipHostFree(F->tab_pinned[k]); }
    delete F;
}

extern "C" int dv_batch_track_enqueue(dv_batch* B, const dv_track_job* jobs, int n) {
    if (!B || (n > 0 && !jobs) || n < 0) { dv_set_error(nullptr, "dv_batch_track_enqueue: bad arguments"); return -1; }
    const std::vector<dv_ctx*>& mem = be_batch_members(B);
    if (n == 0 || mem.empty()) return 0;
    dv_ctx* ctx = mem[0];                                       // errors of the shared part are reported on the first member (and the global slot)
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    DvFrontBatch*& Fp = be_batch_front(B);
    if (!Fp) {
        Fp = new DvFrontBatch();
        DV_CHECK((std::getenv("DVINS_CU_PARTITION_FRONT") || std::getenv("DVINS_SOLVE_CUS")) ? dv_group_stream_create(&Fp->stream, be_batch_index(B)) : hipStreamCreateWithFlags(&Fp->stream, hipStreamNonBlocking));
        DV_CHECK(hipEventCreateWithFlags(&Fp->done, hipEventDisableTiming));
        for (int k = 0; k < 2; ++k) DV_CHECK(hipEventCreateWithFlags(&Fp->ev_copy[k], hipEventDisableTiming));
    }
    DvFrontBatch& F = *Fp;
    // ---- which jobs can share launches: raw mode, gray frames, no mask, no undistortion maps, no object tracker, the group's common geometry ----
    std::vector<int> M;                                         // job indices that are batched
    const dv_config* ref = nullptr;
    for (int i = 0; i < n; ++i) {
        const dv_track_job& j = jobs[i];
        if (j.member < 0 || j.member >= (int)mem.size()) DV_FAIL("dv_batch_track_enqueue: member index out of range");
        dv_ctx* c = mem[j.member];
        for (int q = 0; q < i; ++q) if (jobs[q].member == j.member) DV_FAIL("dv_batch_track_enqueue: a member appears twice");
        const bool plain = j.mode == DV_MODE_RAW && !j.mask && !(j.mem & DV_FMT_BGR) && !c->undist[0] && !c->inst && !c->timing && j.gray0 && (!c->cfg.stereo || j.gray1);
        const bool same = !ref || (c->cfg.width == ref->width && c->cfg.height == ref->height && c->cfg.stereo == ref->stereo && c->cfg.flow_back == ref->flow_back);
        if (plain && same) { if (!ref) ref = &c->cfg; M.push_back(i); }
        else {      // its own launches on its own stream (ordered behind whatever the batch stream still holds for it: dv_track_stereo_enqueue)
            if (dv_track_stereo_enqueue(c, j.gray0, j.gray1, c->cfg.width, c->cfg.height, j.stride > 0 ? j.stride : c->cfg.width * ((j.mem & DV_FMT_BGR) ? 3 : 1), j.t, j.mask, j.mode, j.mem)) { dv_set_error(ctx, c->err); return -1; }
            F.members_single++;
        }
    }
    const int S = (int)M.size();
    if (S == 0) return 0;
    if (S == 1) {      // nothing to share
        const dv_track_job& j = jobs[M[0]]; dv_ctx* c = mem[j.member];
        if (dv_track_stereo_enqueue(c, j.gray0, j.gray1, c->cfg.width, c->cfg.height, j.stride > 0 ? j.stride : c->cfg.width, j.t, nullptr, DV_MODE_RAW, j.mem)) { dv_set_error(ctx, c->err); return -1; }
        F.members_single++;
        return 0;
    }
    const int w = ref->width, h = ref->height; const bool stereo = ref->stereo != 0;
    hipStream_t s = F.stream;
    // ---- per member: the checks and the lazily created resources of dv_track_stereo_enqueue ----
    // (a member's `cur` flips here because everything below addresses its pyramids through it; a failure further down — table growth, the upload, an LDS attribute — flips it
    //  back, so that no member is left with the wrong current pyramid and no pending frame: ADVICE r4)
    struct CurGuard { std::vector<dv_ctx*> flipped; bool committed = false; ~CurGuard() { if (!committed) for (dv_ctx* c : flipped) c->cur ^= 1; } } cur_guard;
    int n_max = 0;
    for (int i : M) {
        const dv_track_job& j = jobs[i]; dv_ctx* c = mem[j.member];
        if (c->pending) { dv_set_error(ctx, "dv_batch_track_enqueue: a member's previous frame was not collected"); return -1; }
        if (ensure_hw(c, c->cfg.min_dist) || ensure_cand(c, w, h)) { dv_set_error(ctx, c->err); return -1; }
        if (c->last_front && c->last_front != s && c->last_done) DV_CHECK(hipStreamWaitEvent(s, c->last_done, 0));      // its previous frame ran on its own stream
        c->cur ^= 1; cur_guard.flipped.push_back(c);
        DV_CHECK(c->left[c->cur].alloc(w, h, 3));
        if (stereo) DV_CHECK(c->right.alloc(w, h, 3));
        n_max = std::max(n_max, c->cfg.max_cnt);
    }
    // ---- the job tables of the round: one pinned block, one upload ----
    const size_t o_pyr = 0, o_apr = o_pyr + (size_t)3 * S * sizeof(DvPyrJob), o_lk = (o_apr + (size_t)2 * S * sizeof(DvPyr) + 255) / 256 * 256,
                 o_cmp = (o_lk + (size_t)2 * S * sizeof(DvLkJob) + 255) / 256 * 256, o_gt = (o_cmp + (size_t)S * sizeof(DvCompactJob) + 255) / 256 * 256,
                 o_gs = (o_gt + (size_t)S * sizeof(GfttTileArgs) + 255) / 256 * 256, o_fin = (o_gs + (size_t)S * sizeof(GfttSelectArgs) + 255) / 256 * 256,
                 total = (o_fin + (size_t)S * sizeof(DvFinalizeJob) + 255) / 256 * 256;
    const int par = F.parity; F.parity ^= 1;
    if (F.copy_used[par]) DV_CHECK(hipEventSynchronize(F.ev_copy[par]));      // the upload that last read this pinned block (two rounds ago) has run
    if (F.tab_bytes < total) {
        DV_CHECK(hipStreamSynchronize(s));
        const size_t cap = total * 2;
        for (int k = 0; k < 2; ++k) {
            DV_CHECK(F.tab[k].ensure(cap));
            if (F.tab_pinned[k]) (void)hipHostFree(F.tab_pinned[k]);
            F.tab_pinned[k] = nullptr;
            DV_CHECK(hipHostMalloc(&F.tab_pinned[k], cap, hipHostMallocDefault));
        }
        F.tab_bytes = cap;
    }
    uint8_t* hp = (uint8_t*)F.tab_pinned[par]; const uint8_t* dp = (const uint8_t*)F.tab[par].p;
    DvPyrJob* h_pyr = (DvPyrJob*)(hp + o_pyr); DvPyr* h_apr = (DvPyr*)(hp + o_apr); DvLkJob* h_lk = (DvLkJob*)(hp + o_lk); DvCompactJob* h_cmp = (DvCompactJob*)(hp + o_cmp);
    GfttTileArgs* h_gt = (GfttTileArgs*)(hp + o_gt); GfttSelectArgs* h_gs = (GfttSelectArgs*)(hp + o_gs); DvFinalizeJob* h_fin = (DvFinalizeJob*)(hp + o_fin);
    int levels = 0, lw[DV_MAX_LEVELS] = { 0 }, lh[DV_MAX_LEVELS] = { 0 };
    for (int k = 0; k < S; ++k) {
        const dv_track_job& j = jobs[M[k]]; dv_ctx* c = mem[j.member];
        const dv_config& cf = c->cfg;
        PyrSet& L = c->left[c->cur]; PyrSet& Lp = c->left[c->cur ^ 1];
        const DvPyr& a = L.pyr; const DvPyr* b = stereo ? &c->right.pyr : nullptr;
        const int stride = j.stride > 0 ? j.stride : w;
        const bool dev = j.mem == DV_MEM_DEVICE || j.mem == DV_MEM_PINNED;      // (pinned + mapped host memory: read in place by the level-1 kernel, like HBM)
        if (!dev || a.levels == 1) {      // host frames: the upload IS the level-0 copy
            DV_CHECK(hipMemcpy2DAsync(a.L[0].p, a.L[0].pitch, j.gray0, stride, w, h, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
            if (b) DV_CHECK(hipMemcpy2DAsync(b->L[0].p, b->L[0].pitch, j.gray1, stride, w, h, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        }
        levels = a.levels;
        for (int l = 1; l < a.levels; ++l) {
            const bool fuse = dev && l == 1;
            DvPyrJob pj{};
            pj.src0 = fuse ? j.gray0 : a.L[l - 1].p; pj.src1 = b ? (fuse ? j.gray1 : b->L[l - 1].p) : nullptr;
            pj.dst0 = a.L[l].p; pj.dst1 = b ? b->L[l].p : nullptr;
            pj.sw = a.L[l - 1].w; pj.sh = a.L[l - 1].h; pj.spitch = fuse ? stride : a.L[l - 1].pitch; pj.dw = a.L[l].w; pj.dh = a.L[l].h; pj.dpitch = a.L[l].pitch;
            pj.cpy0 = fuse ? a.L[0].p : nullptr; pj.cpy1 = (fuse && b) ? b->L[0].p : nullptr; pj.cpitch = a.L[0].pitch;
            h_pyr[(size_t)(l - 1) * S + k] = pj;
            lw[l] = a.L[l].w; lh[l] = a.L[l].h;
        }
        h_apr[2 * k] = a; h_apr[2 * k + 1] = b ? *b : a;      // (mono: the second entry repeats the first — idempotent)
        // temporal LK (skipped by its own n_feat == 0 on a sequence's first frame), stereo LK
        DvLkJob t{}; t.A = Lp.pyr; t.B = a; t.pts_a = c->tr.last_pts; t.n_dev = c->tr.n_feat; t.pts_b = c->tr.lk_pts; t.status = c->tr.lk_status;
        if (!c->have_prev) t.A = a;                           // (no previous pyramid yet: n_feat is 0, nothing is read)
        h_lk[k] = t;
        DvLkJob r{}; r.A = a; r.B = b ? *b : a; r.pts_a = c->tr.curr_pts; r.n_dev = c->tr.n_feat; r.pts_b = c->tr.right_pts; r.status = c->tr.right_status;
        h_lk[S + k] = r;
        h_cmp[k] = DvCompactJob{ c->tr, nullptr, 0, 1, c->n_cand, c->max_ord };
        GfttTileArgs g{};
        g.img = a.L[0].p; g.w = w; g.h = h; g.pitch = a.L[0].pitch; g.in_mask = nullptr; g.mask_pitch = 0;
        g.disc_pts = c->tr.curr_pts; g.n_disc = c->tr.n_tracked; g.radius = cf.min_dist; g.hw = (const uint8_t*)c->hw_buf.p;
        g.n_feat = c->tr.n_feat; g.max_cnt = cf.max_cnt; g.min_new = 1; g.eig_out = nullptr; g.eig_pitch = 0;
        g.cand = (DvCand*)c->cand_buf.p; g.cand_cap = c->cand_cap; g.n_cand = c->n_cand; g.max_ord = c->max_ord;
        h_gt[k] = g;
        GfttSelectArgs q{};
        q.cand = (const DvCand*)c->cand_buf.p; q.n_cand = c->n_cand; q.cand_cap = c->cand_cap; q.max_ord = c->max_ord;
        q.w = w; q.h = h; q.quality = 0.01; q.min_dist = (double)cf.min_dist; q.max_n_host = 0; q.n_feat = c->tr.n_feat; q.max_cnt = cf.max_cnt; q.min_new = 1;
        q.out_xy = nullptr; q.n_out = nullptr; q.tr = c->tr; q.has_tr = 1; q.err_flag = c->err_flag;
        h_gs[k] = q;
        DvFinalizeJob f{};
        f.tr = c->tr; f.cam0 = cf.cam0; f.cam1 = cf.cam1; f.stereo = stereo ? 1 : 0; f.dt = j.t - c->prev_time; f.out = c->out_pinned; f.n_out = c->nout_pinned;
        f.err_in = c->err_flag; f.err_out = c->err_pinned;
        h_fin[k] = f;
    }
    DV_CHECK(dv_copy_async(F.tab[par].p, hp, total, s));
    DV_CHECK(hipEventRecord(F.ev_copy[par], s)); F.copy_used[par] = true;
    // ---- the stages ----
    for (int l = 1; l < levels; ++l) dv_launch_pyr_down_multi((const DvPyrJob*)(dp + o_pyr) + (size_t)(l - 1) * S, S, lw[l], lh[l], s);
    dv_launch_pyr_apron_multi((const DvPyr*)(dp + o_apr), 2 * S, levels, s);
    dv_launch_lk_track_multi((const DvLkJob*)(dp + o_lk), S, n_max, ref->flow_back, 0.5f, s);
    dv_launch_compact_multi((const DvCompactJob*)(dp + o_cmp), S, s);
    dv_launch_gftt_tile_multi((const GfttTileArgs*)(dp + o_gt), S, w, h, s);
    if (dv_launch_gftt_select_multi((const GfttSelectArgs*)(dp + o_gs), S, s)) DV_FAIL("gftt_select: cannot set dynamic LDS size");
    if (stereo) dv_launch_lk_track_multi((const DvLkJob*)(dp + o_lk) + S, S, n_max, ref->flow_back, 0.5f, s);
    dv_launch_finalize_multi((const DvFinalizeJob*)(dp + o_fin), S, n_max, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipEventRecord(F.done, s));
    for (int i : M) {
        const dv_track_job& j = jobs[i]; dv_ctx* c = mem[j.member];
        c->last_done = F.done; c->last_front = s;
        c->prev_time = j.t; c->have_prev = true; c->pending = true;
    }
    cur_guard.committed = true;
    F.rounds++; F.members_batched += S;
    return 0;
}
extern "C" int dv_batch_track_info(dv_batch* B, long long* rounds, long long* members_batched, long long* members_single) {
    if (!B) return -1;
    DvFrontBatch* F = be_batch_front(B);
    if (rounds) *rounds = F ? F->rounds : 0;
    if (members_batched) *members_batched = F ? F->members_batched : 0;
    if (members_single) *members_single = F ? F->members_single : 0;
    return 0;
}

extern "C" {

// ------------------------------- operator-level entries -------------------------------

// copies `bytes` from user memory (host or device) into a ctx scratch buffer on the device
static int stage_in(dv_ctx* ctx, DevBuf& b, const void* src, size_t bytes, int mem) {
    DV_CHECK(b.ensure(bytes ? bytes : 1));
    if (bytes) DV_CHECK(hipMemcpyAsync(b.p, src, bytes, mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
static int stage_out(dv_ctx* ctx, void* dst, const void* src, size_t bytes, int mem) {
    if (bytes) DV_CHECK(hipMemcpyAsync(dst, src, bytes, mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int dv_lk(dv_ctx* ctx, const uint8_t* img_a, const uint8_t* img_b, int w, int h, int stride, const float* pts_a, int n, int max_level,
          int iters, double eps, int use_initial, float* pts_b, uint8_t* status, int mem) {
    if (!ctx) return -1;
    if (!img_a || !img_b || !pts_a || n <= 0) DV_FAIL("dv_lk: empty input (reference throws std::runtime_error, feature_utils.cpp:39-41)");
    if (max_level < 0 || max_level > 3) DV_FAIL("dv_lk: max_level must be in [0,3]");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (build_pyramids(ctx, ctx->opA, nullptr, img_a, nullptr, w, h, stride, mem, max_level)) return -1;
    if (build_pyramids(ctx, ctx->opB, nullptr, img_b, nullptr, w, h, stride, mem, max_level)) return -1;
    const int ml = std::min(ctx->opA.pyr.levels, ctx->opB.pyr.levels) - 1;
    if (stage_in(ctx, ctx->s0, pts_a, (size_t)n * 8, mem)) return -1;
    if (use_initial) { if (stage_in(ctx, ctx->s1, pts_b, (size_t)n * 8, mem)) return -1; }
    else DV_CHECK(ctx->s1.ensure((size_t)n * 8));
    DV_CHECK(ctx->s2.ensure(n));
    iters = std::min(std::max(iters, 0), 100);
    eps = std::min(std::max(eps, 0.), 10.);
    dv_launch_lk_generic(ctx->opA.pyr, ctx->opB.pyr, (const float2*)ctx->s0.p, n, ml, iters, eps * eps, use_initial, (float2*)ctx->s1.p,
                         (uint8_t*)ctx->s2.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (stage_out(ctx, pts_b, ctx->s1.p, (size_t)n * 8, mem)) return -1;
    if (stage_out(ctx, status, ctx->s2.p, n, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_track_by_lk(dv_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int w, int h, int stride, const float* pts1, int n,
                   int flow_back, float dist_thresh, float* pts2, uint8_t* status, int mem) {
    if (!ctx) return -1;
    if (!img1 || !img2 || !pts1 || n <= 0) DV_FAIL("dv_track_by_lk: FeatureTrackByLK() input wrong, received at least one of parameter are empty");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (build_pyramids(ctx, ctx->opA, &ctx->opB, img1, img2, w, h, stride, mem, 3)) return -1;
    if (stage_in(ctx, ctx->s0, pts1, (size_t)n * 8, mem)) return -1;
    DV_CHECK(ctx->s1.ensure((size_t)n * 8));
    DV_CHECK(ctx->s2.ensure(n));
    StageScope sc(ctx, "op_lk_track");
    dv_launch_lk_track(ctx->opA.pyr, ctx->opB.pyr, (const float2*)ctx->s0.p, nullptr, n, flow_back, dist_thresh, (float2*)ctx->s1.p,
                       (uint8_t*)ctx->s2.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (stage_out(ctx, pts2, ctx->s1.p, (size_t)n * 8, mem)) return -1;
    if (stage_out(ctx, status, ctx->s2.p, n, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// cv::cuda::SparsePyrLKOpticalFlow::calc (use_initial: pts_b holds the initial flow) and FeatureTrackByLKGpu, operator forms for the parity tests
static int lk_cuda_prepare(dv_ctx* ctx, const uint8_t* img_a, const uint8_t* img_b, int w, int h, int stride, int mem, int max_level) {
    if (build_pyramids(ctx, ctx->opA, &ctx->opB, img_a, img_b, w, h, stride, mem, 0)) return -1;          // level 0 only (pitched copies)
    return build_cuda_pyramids(ctx, ctx->leftc[0], &ctx->leftc[1], ctx->opA.pyr, &ctx->opB.pyr, w, h, max_level);
}
int dv_lk_cuda(dv_ctx* ctx, const uint8_t* img_a, const uint8_t* img_b, int w, int h, int stride, const float* pts_a, int n, int max_level, int iters, int use_initial,
               float* pts_b, uint8_t* status, int mem) {
    if (!ctx) return -1;
    if (!img_a || !img_b || !pts_a || n <= 0) DV_FAIL("dv_lk_cuda: empty input");
    if (max_level < 0 || max_level > 3 || iters < 0 || iters > 100) DV_FAIL("dv_lk_cuda: max_level must be in [0,3], iters in [0,100]");
    if (ctx->pending || ctx->have_prev) DV_FAIL("dv_lk_cuda: operator calls need a ctx that is not tracking a sequence (its pyramids are used as scratch)");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (lk_cuda_prepare(ctx, img_a, img_b, w, h, stride, mem, max_level)) return -1;
    const int ml = std::min(ctx->leftc[0].pyr.levels, ctx->leftc[1].pyr.levels) - 1;
    if (stage_in(ctx, ctx->s0, pts_a, (size_t)n * 8, mem)) return -1;
    if (use_initial) { if (stage_in(ctx, ctx->s1, pts_b, (size_t)n * 8, mem)) return -1; } else DV_CHECK(ctx->s1.ensure((size_t)n * 8));
    DV_CHECK(ctx->s2.ensure(n));
    dv_launch_lk_cuda_generic(ctx->leftc[0].pyr, ctx->leftc[1].pyr, (const float2*)ctx->s0.p, n, ml, iters, use_initial, (float2*)ctx->s1.p, (uint8_t*)ctx->s2.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (stage_out(ctx, pts_b, ctx->s1.p, (size_t)n * 8, mem)) return -1;
    if (stage_out(ctx, status, ctx->s2.p, n, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
int dv_track_by_lk_gpu(dv_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int w, int h, int stride, const float* pts1, int n, int flow_back, float* pts2, uint8_t* status, int mem) {
    if (!ctx) return -1;
    if (!img1 || !img2 || !pts1 || n <= 0) DV_FAIL("dv_track_by_lk_gpu: flowTrack() input wrong, received at least one of parameter are empty");
    if (ctx->pending || ctx->have_prev) DV_FAIL("dv_track_by_lk_gpu: operator calls need a ctx that is not tracking a sequence (its pyramids are used as scratch)");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (lk_cuda_prepare(ctx, img1, img2, w, h, stride, mem, 3)) return -1;
    if (stage_in(ctx, ctx->s0, pts1, (size_t)n * 8, mem)) return -1;
    DV_CHECK(ctx->s1.ensure((size_t)n * 8)); DV_CHECK(ctx->s2.ensure(n));
    dv_launch_lk_cuda_track(ctx->leftc[0].pyr, ctx->leftc[1].pyr, (const float2*)ctx->s0.p, nullptr, n, flow_back, 1.0f, (float2*)ctx->s1.p, (uint8_t*)ctx->s2.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (stage_out(ctx, pts2, ctx->s1.p, (size_t)n * 8, mem)) return -1;
    if (stage_out(ctx, status, ctx->s2.p, n, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}
// stage an image (host or device) into a pitched device buffer; returns pointer/pitch to use
static int stage_image(dv_ctx* ctx, DevBuf& b, const uint8_t* img, int w, int h, int stride, int mem, const uint8_t** out, int* pitch) {
    if (mem == DV_MEM_DEVICE) { *out = img; *pitch = stride; return 0; }
    const int p = align_up(w, 16);
    DV_CHECK(b.ensure((size_t)p * h + 64));
    DV_CHECK(hipMemcpy2DAsync(b.p, p, img, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
    *out = (const uint8_t*)b.p; *pitch = p;
    return 0;
}

// cuda::pyrDown on an 8-bit image (the level step of that tracker's pyramid); dst is ((w+1)/2) x ((h+1)/2), tightly packed
int dv_pyr_down_cuda(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int mem) {
    if (!ctx) return -1;
    if (!src || !dst) DV_FAIL("dv_pyr_down_cuda: null argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const uint8_t* d_src; int pitch;
    if (stage_image(ctx, ctx->s0, src, w, h, stride, mem, &d_src, &pitch)) return -1;
    const int dw = (w + 1) / 2, dh = (h + 1) / 2, dp = align_up(dw, 16);
    DV_CHECK(ctx->s1.ensure((size_t)dp * dh + 64));
    dv_launch_pyr_down2(d_src, nullptr, w, h, pitch, (uint8_t*)ctx->s1.p, nullptr, dp, nullptr, nullptr, 0, ctx->stream, 1);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(dst, dw, ctx->s1.p, dp, dw, dh, mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

static int min_eigen_rule(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem, int rule) {
    if (!ctx) return -1;
    if (!img || !eig) DV_FAIL("dv_min_eigen: null argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const uint8_t* d_img; int pitch;
    if (stage_image(ctx, ctx->s0, img, w, h, stride, mem, &d_img, &pitch)) return -1;
    if (ensure_cand(ctx, w, h)) return -1;
    float* d_eig = eig;
    if (mem != DV_MEM_DEVICE) { DV_CHECK(ctx->s1.ensure((size_t)w * h * 4)); d_eig = (float*)ctx->s1.p; }
    DV_CHECK(hipMemsetAsync(ctx->n_cand, 0, 8, ctx->stream));     // n_cand + max_ord
    GfttTileArgs a{};
    a.img = d_img; a.w = w; a.h = h; a.pitch = pitch; a.eig_out = d_eig; a.eig_pitch = w;
    a.cand = (DvCand*)ctx->cand_buf.p; a.cand_cap = ctx->cand_cap; a.n_cand = ctx->n_cand; a.max_ord = ctx->max_ord; a.rule = rule;
    dv_launch_gftt_tile(a, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (mem != DV_MEM_DEVICE) DV_CHECK(hipMemcpyAsync(eig, d_eig, (size_t)w * h * 4, hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_min_eigen(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem) { return min_eigen_rule(ctx, img, w, h, stride, eig, mem, DV_GFTT_RULE_CPU); }
int dv_min_eigen_cuda(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem) { return min_eigen_rule(ctx, img, w, h, stride, eig, mem, DV_GFTT_RULE_CUDA); }

static int gftt_rule_op(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride, int max_n, double quality,
                        double min_dist, float* out_xy, int* n_out, int mem, int rule) {
    if (!ctx) return -1;
    if (!img || !out_xy || !n_out) DV_FAIL("dv_gftt: null argument");
    if (!(quality > 0)) DV_FAIL("dv_gftt: qualityLevel must be > 0");
    if (min_dist < 0) DV_FAIL("dv_gftt: minDistance must be >= 0");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const uint8_t *d_img, *d_mask = nullptr; int pitch, mpitch = 0;
    if (stage_image(ctx, ctx->s0, img, w, h, stride, mem, &d_img, &pitch)) return -1;
    if (mask_or_null && stage_image(ctx, ctx->s3, mask_or_null, w, h, stride, mem, &d_mask, &mpitch)) return -1;
    if (ensure_cand(ctx, w, h)) return -1;
    DV_CHECK(ctx->s1.ensure((size_t)DV_MAX_FEATS * 8 + 16));
    float2* d_out = (float2*)ctx->s1.p; int* d_n = (int*)((uint8_t*)ctx->s1.p + (size_t)DV_MAX_FEATS * 8);
    DV_CHECK(hipMemsetAsync(ctx->n_cand, 0, 12, ctx->stream));    // n_cand, max_ord, err_flag
    GfttTileArgs a{};
    a.img = d_img; a.w = w; a.h = h; a.pitch = pitch; a.in_mask = d_mask; a.mask_pitch = mpitch;
    a.cand = (DvCand*)ctx->cand_buf.p; a.cand_cap = ctx->cand_cap; a.n_cand = ctx->n_cand; a.max_ord = ctx->max_ord; a.rule = rule;
    dv_launch_gftt_tile(a, ctx->stream);
    GfttSelectArgs sa{};
    sa.cand = (const DvCand*)ctx->cand_buf.p; sa.n_cand = ctx->n_cand; sa.cand_cap = ctx->cand_cap; sa.max_ord = ctx->max_ord;
    sa.w = w; sa.h = h; sa.quality = quality; sa.min_dist = min_dist; sa.max_n_host = max_n; sa.n_feat = nullptr;
    sa.out_xy = d_out; sa.n_out = d_n; sa.has_tr = 0; sa.err_flag = ctx->err_flag; sa.rule = rule;
    if (dv_launch_gftt_select(sa, ctx->stream)) DV_FAIL("gftt_select: cannot set dynamic LDS size");
    DV_CHECK(hipGetLastError());
    int n = 0, ef = 0;
    DV_CHECK(hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipMemcpyAsync(&ef, ctx->err_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    if (ef) { DV_CHECK(hipMemsetAsync(ctx->err_flag, 0, 4, ctx->stream)); DV_FAIL("dv_gftt: device error flags=" + std::to_string(ef)); }
    if (stage_out(ctx, out_xy, d_out, (size_t)n * 8, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    *n_out = n;
    return 0;
}

int dv_gftt(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride, int max_n, double quality,
            double min_dist, float* out_xy, int* n_out, int mem) {
    return gftt_rule_op(ctx, img, mask_or_null, w, h, stride, max_n, quality, min_dist, out_xy, n_out, mem, DV_GFTT_RULE_CPU);
}
int dv_gftt_cuda(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride, int max_n, double quality,
                 double min_dist, float* out_xy, int* n_out, int mem) {
    return gftt_rule_op(ctx, img, mask_or_null, w, h, stride, max_n, quality, min_dist, out_xy, n_out, mem, DV_GFTT_RULE_CUDA);
}

int dv_viode_mask(dv_ctx* ctx, const uint8_t* seg_bgr, int w, int h, int stride, const uint32_t* dyn_keys, int nkeys, uint8_t* merge_mask, uint8_t* inv_merge_mask,
                  uint32_t* key_image, int32_t* boxes) {
    if (!ctx) return -1;
    if (!seg_bgr || !dyn_keys || !merge_mask || !inv_merge_mask || !boxes || w <= 0 || h <= 0 || stride < 3 * w) DV_FAIL("dv_viode_mask: bad argument");
    if (nkeys < 1 || nkeys > 64) DV_FAIL("dv_viode_mask: 1..64 dynamic keys");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    const int sp = align_up(3 * w, 16), mp = align_up(w, 16);
    DV_CHECK(ctx->s0.ensure((size_t)sp * h));                             // label image
    DV_CHECK(ctx->s1.ensure((size_t)mp * h)); DV_CHECK(ctx->s2.ensure((size_t)mp * h));      // merge, inverse
    DV_CHECK(ctx->s3.ensure((size_t)4 * w * h + 4096));                    // key image | keys | boxes
    DV_CHECK(ctx->s4.ensure(4096));
    uint32_t* d_keys = (uint32_t*)ctx->s4.p; int32_t* d_box = (int32_t*)((uint8_t*)ctx->s4.p + 1024);
    int32_t init[256];
    for (int k = 0; k < 64; ++k) { init[4 * k] = 0x7fffffff; init[4 * k + 1] = -1; init[4 * k + 2] = 0x7fffffff; init[4 * k + 3] = -1; }
    DV_CHECK(hipMemcpy2DAsync(ctx->s0.p, sp, seg_bgr, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(d_keys, dyn_keys, 4 * (size_t)nkeys, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync(d_box, init, 16 * (size_t)nkeys, hipMemcpyHostToDevice, s));
    DV_CHECK(hipStreamSynchronize(s));                                     // init[] is a stack object
    dv_launch_viode_mask((const uint8_t*)ctx->s0.p, w, h, sp, d_keys, nkeys, (uint8_t*)ctx->s1.p, (uint8_t*)ctx->s2.p, mp, key_image ? (uint32_t*)ctx->s3.p : nullptr, d_box, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(merge_mask, w, ctx->s1.p, mp, w, h, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpy2DAsync(inv_merge_mask, w, ctx->s2.p, mp, w, h, hipMemcpyDeviceToHost, s));
    if (key_image) DV_CHECK(hipMemcpyAsync(key_image, ctx->s3.p, (size_t)4 * w * h, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(boxes, d_box, 16 * (size_t)nkeys, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    for (int k = 0; k < nkeys; ++k) if (boxes[4 * k + 1] < 0) { boxes[4 * k] = boxes[4 * k + 2] = -1; boxes[4 * k + 3] = -1; }      // key not present
    return 0;
}

int dv_bgr2gray(dv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int mem) {
    if (!ctx) return -1;
    if (!bgr || !gray || w <= 0 || h <= 0 || stride < 3 * w) DV_FAIL("dv_bgr2gray: bad argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    const uint8_t* src = bgr; int sp = stride;
    if (mem != DV_MEM_DEVICE) {
        sp = align_up(3 * w, 16);
        DV_CHECK(ctx->s3.ensure((size_t)sp * h));
        DV_CHECK(hipMemcpy2DAsync(ctx->s3.p, sp, bgr, stride, (size_t)3 * w, h, hipMemcpyHostToDevice, s));
        src = (const uint8_t*)ctx->s3.p;
    }
    const int dp = align_up(w, 16);
    DV_CHECK(ctx->s4.ensure((size_t)dp * h));
    dv_launch_bgr2gray(src, nullptr, w, h, sp, (uint8_t*)ctx->s4.p, nullptr, dp, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(gray, w, ctx->s4.p, dp, w, h, mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    return 0;
}

int dv_remap(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, int channels, const int16_t* map1_xy, const uint16_t* map2, uint8_t* dst, int mem) {
    if (!ctx) return -1;
    if (!src || !dst || !map1_xy || !map2 || w <= 0 || h <= 0 || (channels != 1 && channels != 3) || stride < channels * w) DV_FAIL("dv_remap: bad argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    const bool dev = mem == DV_MEM_DEVICE;
    const uint8_t* d_src = src; int sp = stride;
    if (!dev) {
        sp = align_up(channels * w, 16);
        DV_CHECK(ctx->s3.ensure((size_t)sp * h));
        DV_CHECK(hipMemcpy2DAsync(ctx->s3.p, sp, src, stride, (size_t)channels * w, h, hipMemcpyHostToDevice, s));
        d_src = (const uint8_t*)ctx->s3.p;
    }
    const size_t npx = (size_t)w * h;
    DV_CHECK(ctx->s2.ensure(6 * npx));
    DV_CHECK(hipMemcpyAsync(ctx->s2.p, map1_xy, 4 * npx, hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemcpyAsync((uint8_t*)ctx->s2.p + 4 * npx, map2, 2 * npx, hipMemcpyHostToDevice, s));
    const int dp = align_up(channels * w, 16);
    DV_CHECK(ctx->s4.ensure((size_t)dp * h));
    dv_launch_remap(d_src, nullptr, w, h, sp, channels, 0, (const int16_t*)ctx->s2.p, (const uint16_t*)((uint8_t*)ctx->s2.p + 4 * npx), nullptr, nullptr,
                    (uint8_t*)ctx->s4.p, nullptr, dp, s);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(dst, (size_t)channels * w, ctx->s4.p, dp, (size_t)channels * w, h, dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    return 0;
}

int dv_set_undistort_maps(dv_ctx* ctx, int cam, const int16_t* map1_xy, const uint16_t* map2, int w, int h) {
    if (!ctx) return -1;
    if (cam < 0 || cam > 1) DV_FAIL("dv_set_undistort_maps: cam must be 0 or 1");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (ctx->pending) DV_FAIL("dv_set_undistort_maps: a frame is in flight");
    if (!map1_xy) { ctx->undist[cam] = false; if (cam == 0) ctx->undist[1] = false; return 0; }
    if (!map2 || w != ctx->cfg.width || h != ctx->cfg.height) DV_FAIL("dv_set_undistort_maps: maps must be width x height of the config");
    if (cam == 1 && !ctx->undist[0]) DV_FAIL("dv_set_undistort_maps: install camera 0 first");
    const size_t npx = (size_t)w * h;
    DV_CHECK(ctx->undist_buf[cam].ensure(6 * npx));
    DV_CHECK(hipMemcpyAsync(ctx->undist_buf[cam].p, map1_xy, 4 * npx, hipMemcpyHostToDevice, ctx->stream));
    DV_CHECK(hipMemcpyAsync((uint8_t*)ctx->undist_buf[cam].p + 4 * npx, map2, 2 * npx, hipMemcpyHostToDevice, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->undist[cam] = true; ctx->undist_w = w; ctx->undist_h = h;
    return 0;
}

int dv_pyr_down(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int mem) {
    if (!ctx) return -1;
    if (!src || !dst) DV_FAIL("dv_pyr_down: null argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const uint8_t* d_src; int pitch;
    if (stage_image(ctx, ctx->s0, src, w, h, stride, mem, &d_src, &pitch)) return -1;
    const int dw = (w + 1) / 2, dh = (h + 1) / 2, dp = align_up(dw, 16);
    DV_CHECK(ctx->s1.ensure((size_t)dp * dh + 64));
    dv_launch_pyr_down2(d_src, nullptr, w, h, pitch, (uint8_t*)ctx->s1.p, nullptr, dp, nullptr, nullptr, 0, ctx->stream);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(dst, dw, ctx->s1.p, dp, dw, dh, mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_circle_mask(dv_ctx* ctx, uint8_t* mask, int w, int h, int stride, const float* pts_xy, int n, int radius, int mem) {
    if (!ctx) return -1;
    if (!mask || (n > 0 && !pts_xy)) DV_FAIL("dv_circle_mask: null argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (ensure_hw(ctx, radius)) return -1;
    uint8_t* d_mask = mask; int pitch = stride;
    if (mem != DV_MEM_DEVICE) {
        pitch = align_up(w, 16);
        DV_CHECK(ctx->s0.ensure((size_t)pitch * h));
        DV_CHECK(hipMemcpy2DAsync(ctx->s0.p, pitch, mask, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
        d_mask = (uint8_t*)ctx->s0.p;
    }
    if (stage_in(ctx, ctx->s1, pts_xy, (size_t)n * 8, mem)) return -1;
    dv_launch_circle_mask(d_mask, w, h, pitch, (const float2*)ctx->s1.p, n, radius, (const uint8_t*)ctx->hw_buf.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (mem != DV_MEM_DEVICE) DV_CHECK(hipMemcpy2DAsync(mask, stride, d_mask, pitch, w, h, hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_erode(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, int k, uint8_t* dst, int mem) {
    if (!ctx) return -1;
    if (!src || !dst || k < 1) DV_FAIL("dv_erode: bad argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const uint8_t* d_src; int pitch;
    if (stage_image(ctx, ctx->s0, src, w, h, stride, mem, &d_src, &pitch)) return -1;
    const int p = align_up(w, 16);
    DV_CHECK(ctx->s1.ensure((size_t)p * h)); DV_CHECK(ctx->s2.ensure((size_t)p * h));
    dv_launch_erode(d_src, w, h, pitch, k, (uint8_t*)ctx->s1.p, p, (uint8_t*)ctx->s2.p, p, ctx->stream);
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipMemcpy2DAsync(dst, mem == DV_MEM_DEVICE ? stride : w, ctx->s2.p, p, w, h,
                              mem == DV_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

int dv_lift_projective(dv_ctx* ctx, const dv_cam* cam, const float* pts_xy, int n, float* out_xy, int mem) {
    return dv_lift_projective_offset(ctx, cam, pts_xy, n, 0.0, 0.0, out_xy, mem);
}

int dv_lift_projective_offset(dv_ctx* ctx, const dv_cam* cam, const float* pts_xy, int n, double off_x, double off_y, float* out_xy, int mem) {
    if (!ctx) return -1;
    if (!cam || (n > 0 && (!pts_xy || !out_xy))) DV_FAIL("dv_lift_projective: null argument");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (stage_in(ctx, ctx->s0, pts_xy, (size_t)n * 8, mem)) return -1;
    DV_CHECK(ctx->s1.ensure((size_t)std::max(n, 1) * 8));
    dv_launch_lift(*cam, (const float2*)ctx->s0.p, n, off_x, off_y, (float2*)ctx->s1.p, ctx->stream);
    DV_CHECK(hipGetLastError());
    if (stage_out(ctx, out_xy, ctx->s1.p, (size_t)n * 8, mem)) return -1;
    DV_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// FrameLines::UndistortedLineEndPoints (line_detector, front end of TrackImageLine): both end points through PinholeCamera::liftProjective; the Line keeps
// them as cv::Point2f, LineFeature widens to double (basic/line_feature.h:31-33)
int dv_undistort_lines(dv_ctx* ctx, const dv_cam* cam, const float* lines_xyxy, int n, double* out_xyxy) {
    if (!ctx) return -1;
    if (!cam || n < 0 || (n > 0 && (!lines_xyxy || !out_xyxy))) DV_FAIL("dv_undistort_lines: bad argument");
    if (n == 0) return 0;
    std::vector<float> un((size_t)4 * n);
    if (dv_lift_projective(ctx, cam, lines_xyxy, 2 * n, un.data(), DV_MEM_HOST)) return -1;
    for (size_t i = 0; i < (size_t)4 * n; ++i) out_xyxy[i] = (double)un[i];
    return 0;
}

} // extern "C"
