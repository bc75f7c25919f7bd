// inst_track.hip — the per-object half of the dynamic-mode front end on gfx950: InstsFeatManager of the reference
//   InstsFeatManager::{InstsTrack,ManageInstances,Output,BoxAssociate2Dto3D,AddInstancesByTracking}   front_end/dynamic_tracker.cpp:61-152,348-577,791-828
//   InstFeat::{TrackLeft,TrackRightByPad,UndistortedPointsWithAddOffset,PtsVelocity,RightPtsVelocity,PostProcess}   front_end/instance_feature.cpp, instance_feature.h:88-101
//   InstanceImagePadding                                                                                front_end/feature_utils.cpp:406-413
// Host side (this file, C++ inside the .so): the object table keyed by track id, the lost_num life cycle, the 2-D / 3-D box association.
// Device side: every object owns a DvTrackState (points, ids, track counts, previous undistorted points) and two ROI pyramids in HBM; per frame
// and per visible object the stream carries: ROI crop + zero padding (roi_pad_kernel) -> pyrDown of the padded pair -> lk_track (ROI-local,
// fwd + bwd + InBorder) -> compaction -> 5x5 erosion of the object mask -> Shi-Tomasi (tile + select, discs of the tracked points excluded,
// ids from the tracker-wide counter) -> lk_track into the right image with the box offset added to the points (TrackRightByPad) ->
// finalize (undistortion with the box offset, velocities, PostProcess) -> one D2H copy per frame for all objects.  No host round trip inside
// a frame: the per-object point counts stay on the device.
// Canonical choices: objects are visited in ascending id (the reference iterates an unordered_map) and run after the background tracker on the
// same stream, so the shared id counter hands out ids in a fixed order (the reference increments it from two threads without a lock).
#include <algorithm>
#include <map>
#include "dv_ctx.h"

#define INST_CAP 256          // device capacity per object (max_dynamic_cnt <= 200)

namespace {

__global__ __launch_bounds__(256) void roi_pad_kernel(const uint8_t* __restrict__ src, int spitch, int x0, int y0, int w, int h, uint8_t* __restrict__ dst, int dpitch, int W, int H) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    dst[(size_t)y * dpitch + x] = (x < w && y < h) ? src[(size_t)(y0 + y) * spitch + x0 + x] : (uint8_t)0;      // InstanceImagePadding: zero fill right / below
}

// the same for several crops in ONE launch (blockIdx.z = job): the crops and re-paddings of all objects of a frame
struct RoiJob { const uint8_t* src; uint8_t* dst; int spitch, x0, y0, w, h, dpitch, W, H; };
__global__ __launch_bounds__(256) void roi_pad_multi_kernel(const RoiJob* __restrict__ jobs) {
    const RoiJob j = jobs[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= j.W || y >= j.H) return;
    j.dst[(size_t)y * j.dpitch + x] = (x < j.w && y < j.h) ? j.src[(size_t)(j.y0 + y) * j.spitch + j.x0 + x] : (uint8_t)0;
}

struct RoiPyr {          // pyramid storage with a fixed capacity (the full frame), re-laid-out per frame without re-allocation
    DevBuf buf; DvPyr pyr{}; int w = 0, h = 0;
    hipError_t reserve(int W, int H) {
        size_t total = 0; int cw = W, ch = H;
        for (int l = 0; l < DV_MAX_LEVELS; ++l) { total += (size_t)align_up(cw, 16) * ch; total = (total + 255) / 256 * 256; cw = (cw + 1) / 2; ch = (ch + 1) / 2; }
        return buf.ensure(total + 512);
    }
    void layout(int w_, int h_) {      // buildOpticalFlowPyramid's level rule, as PyrSet::alloc
        w = w_; h = h_;
        int n = 0, cw = w, ch = h; size_t off = 0;
        for (int l = 0; l < DV_MAX_LEVELS; ++l) {
            pyr.L[n] = DvLevel{ (uint8_t*)buf.p + off, cw, ch, align_up(cw, 16), 0 };
            off += (size_t)align_up(cw, 16) * ch; off = (off + 255) / 256 * 256; ++n;
            const int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
            if (nw <= DV_LK_WIN || nh <= DV_LK_WIN) break;
            cw = nw; ch = nh;
        }
        for (int l = n; l < DV_MAX_LEVELS; ++l) pyr.L[l] = DvLevel{ nullptr, 0, 0, 0, 0 };
        pyr.levels = n;
    }
};

struct Slot {            // InstFeat
    unsigned id = 0; int lost_num = 0; bool visible = false, has_box2d = false, has_box3d = false, has_prev = false;
    int class_id = 0; int rx = 0, ry = 0, rw = 0, rh = 0;      // Box2D::rect of the current frame
    int pw = 0, ph = 0;                                        // size of prev_roi_gray
    dv_box3d box3d{};
    DevBuf state; DvTrackState tr{}; int* scal = nullptr;      // n_feat, n_tracked
    RoiPyr roi[2]; int cur = 0;                                // roi[cur]: this frame's (padded) ROI pyramid; roi[cur ^ 1]: level 0 holds prev_roi_gray in its top-left pw x ph
    RoiPyr padA;                                               // previous ROI padded to the common size
    DevBuf tmp, ero, cand; uint8_t* mask = nullptr;            // mask: this frame's object mask inside the tracker's mask area (one upload per frame for all objects)
    int* n_cand = nullptr; unsigned* max_ord = nullptr; int* id_slot = nullptr;      // the object's own Shi-Tomasi scratch: the objects' detections share launches
    const double* points = nullptr; int n_points = 0;          // extra 3-D points of this frame (pass-through)
    std::vector<double> pts_copy;
    bool out_valid = false; int out_index = -1;
    void release() { state.release(); roi[0].buf.release(); roi[1].buf.release(); padA.buf.release(); tmp.release(); ero.release(); cand.release(); }
};

}  // namespace

struct dv_inst_tracker {
    std::map<unsigned, Slot> slots;
    int max_cnt = 50, min_dist = 5, use_det3d = 0;
    DevBuf hw; int hw_radius = -1;
    double last_time = 0, cur_time = 0;
    void* pinned_in = nullptr; size_t pinned_in_bytes = 0;
    dv_feat* out_dev = nullptr; DevBuf out_buf; void* out_pinned = nullptr; size_t out_cap_slots = 0;      // [slot][INST_CAP] rows + counts
    hipEvent_t done = nullptr; bool pending = false; bool frame_enqueued = false;
    hipStream_t stream = nullptr;                               // the objects run beside the background tracker: own stream, own Shi-Tomasi scratch
    DevBuf jobs; void* jobs_pinned = nullptr; size_t jobs_cap = 0;     // DvLkJob tables of the two batched LK stages (temporal | right)
    DevBuf arena; void* arena_pinned = nullptr; size_t arena_cap = 0;  // per-frame job tables of the batched per-object stages (ROI crops, pyramid levels): filled on the host, ONE upload
    DevBuf scal, mask_all; int cand_cap = 0; int* err_flag = nullptr;
    std::vector<unsigned> out_order;                            // ids written this frame, in output order
    // extra points from the frame's disparity map (dv_inst_set_disparity; extra_points.hip): the reference's second thread = a side stream
    const float* disp_user = nullptr; int disp_stride = 0, disp_mem = 0; double disp_baseline = 0; bool disp_next = false, xp_frame = false, xp_inflight = false;
    const uint32_t* keys_user = nullptr; int keys_stride = 0, keys_mem = 0; bool keys_next = false; DevBuf keys_buf;      // VIODE: seg1's key image of the next frame (dv_inst_set_right_keys)
    DevBuf disp_buf, xp_pool; hipStream_t xstream = nullptr; hipEvent_t ev_xin = nullptr, ev_xdone = nullptr;
    void* xp_pinned = nullptr; size_t xp_cap_slots = 0;          // per output slot: count (64 bytes) + 3 * DV_XP_CAP doubles, written by the kernel straight into pinned memory
    ~dv_inst_tracker() {
        disp_buf.release(); xp_pool.release(); keys_buf.release();
        if (xstream) { (void)hipStreamSynchronize(xstream); (void)hipStreamDestroy(xstream); }
        if (ev_xin) (void)hipEventDestroy(ev_xin);
        if (ev_xdone) (void)hipEventDestroy(ev_xdone);
        if (xp_pinned) (void)hipHostFree(xp_pinned);
        for (auto& kv : slots) kv.second.release();
        hw.release(); out_buf.release(); mask_all.release(); scal.release(); jobs.release();
        if (jobs_pinned) (void)hipHostFree(jobs_pinned);
        arena.release(); if (arena_pinned) (void)hipHostFree(arena_pinned);
        if (stream) (void)hipStreamDestroy(stream);
        if (pinned_in) (void)hipHostFree(pinned_in);
        if (out_pinned) (void)hipHostFree(out_pinned);
        if (done) (void)hipEventDestroy(done);
    }
};

void dv_inst_destroy_internal(dv_inst_tracker* t) { delete t; }
// the background tracker's next frame must not rebuild the right pyramid while the objects of the previous frame still read it
int dv_inst_wait_before_next_frame(dv_ctx* ctx) {
    if (!ctx->inst || !ctx->inst->frame_enqueued) return 0;
    ctx->inst->frame_enqueued = false;
    return hipStreamWaitEvent(ctx->stream, ctx->inst->done, 0) == hipSuccess ? 0 : -1;
}

static void circle_half_widths_i(int radius, std::vector<uint8_t>& hw) {      // cv::circle's midpoint rasteriser (as dvins_api.hip)
    hw.assign(radius + 1, 0);
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hw[dy] = (uint8_t)std::max<int>(hw[dy], dx); hw[dx] = (uint8_t)std::max<int>(hw[dx], dy);
        dy++; err += plus; plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
}

static int slot_init(dv_ctx* ctx, Slot& s) {
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const size_t N = INST_CAP;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_last = take(N * 8), o_cur = take(N * 8), o_lk = take(N * 8), o_lks = take(N), o_ids = take(N * 4), o_cnt = take(N * 4), o_pun = take(N * 8), o_prun = take(N * 8),
                 o_prv = take(N), o_trk = take(N), o_rp = take(N * 8), o_rs = take(N), o_scal = take(64);
    DV_CHECK(s.state.ensure(off));
    DV_CHECK(hipMemsetAsync(s.state.p, 0, off, ctx->inst->stream));
    uint8_t* b = (uint8_t*)s.state.p;
    s.tr.last_pts = (float2*)(b + o_last); s.tr.curr_pts = (float2*)(b + o_cur); s.tr.lk_pts = (float2*)(b + o_lk); s.tr.lk_status = b + o_lks;
    s.tr.ids = (uint32_t*)(b + o_ids); s.tr.track_cnt = (int32_t*)(b + o_cnt); s.tr.prev_un = (float2*)(b + o_pun); s.tr.prev_run = (float2*)(b + o_prun);
    s.tr.prev_rvalid = b + o_prv; s.tr.tracked = b + o_trk; s.tr.right_pts = (float2*)(b + o_rp); s.tr.right_status = b + o_rs;
    s.scal = (int*)(b + o_scal); s.tr.n_feat = s.scal; s.tr.n_tracked = s.scal + 1;
    s.n_cand = s.scal + 2; s.max_ord = (unsigned*)(s.scal + 3); s.id_slot = s.scal + 4;
    s.tr.next_id = ctx->tr.next_id;                      // InstFeat::global_id_count is ONE static counter for background and object features
    for (int k = 0; k < 2; ++k) DV_CHECK(s.roi[k].reserve(W, H));
    DV_CHECK(s.padA.reserve(W, H));
    const size_t mp = (size_t)align_up(W, 16) * H;
    DV_CHECK(s.tmp.ensure(mp)); DV_CHECK(s.ero.ensure(mp));
    DV_CHECK(s.cand.ensure((size_t)ctx->inst->cand_cap * sizeof(DvCand)));
    return 0;
}

static float rect_iou(const float a[4], const float b[4]) {      // Box2D::IoU on (x, y, w, h) rectangles (basic/box2d.cpp:20-27)
    const float x1 = std::max(a[0], b[0]), y1 = std::max(a[1], b[1]), x2 = std::min(a[0] + a[2], b[0] + b[2]), y2 = std::min(a[1] + a[3], b[1] + b[3]);
    const float in = (x2 > x1 && y2 > y1) ? (x2 - x1) * (y2 - y1) : 0.f;
    const float un = a[2] * a[3] + b[2] * b[3] - in;
    if (un < 2.220446049250313e-16) return 0.f;
    return in / un;
}

extern "C" {

int dv_inst_config(dv_ctx* ctx, int max_dynamic_cnt, int min_dynamic_dist, int use_det3d) {
    if (!ctx) return -1;
    if (max_dynamic_cnt < 1 || max_dynamic_cnt > INST_CAP - 8) DV_FAIL("dv_inst_config: max_dynamic_cnt must be in [1, 248]");
    if (min_dynamic_dist < 0 || min_dynamic_dist > DV_MAX_RADIUS) DV_FAIL("dv_inst_config: min_dynamic_dist out of range [0,128]");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    if (!ctx->inst) {
        ctx->inst = new dv_inst_tracker();
        DV_CHECK(hipEventCreateWithFlags(&ctx->inst->done, hipEventDisableTiming));
        DV_CHECK(hipStreamCreateWithFlags(&ctx->inst->stream, hipStreamNonBlocking));
        DV_CHECK(hipStreamCreateWithFlags(&ctx->inst->xstream, hipStreamNonBlocking));
        DV_CHECK(hipEventCreateWithFlags(&ctx->inst->ev_xin, hipEventDisableTiming));
        DV_CHECK(hipEventCreateWithFlags(&ctx->inst->ev_xdone, hipEventDisableTiming));
        const int cap = std::max(4096, (ctx->cfg.width * ctx->cfg.height) / 4);
        ctx->inst->cand_cap = cap;
        DV_CHECK(ctx->inst->scal.ensure(256)); DV_CHECK(hipMemset(ctx->inst->scal.p, 0, 256));
        ctx->inst->err_flag = (int*)ctx->inst->scal.p + 2;
    }
    dv_inst_tracker& T = *ctx->inst;
    if (T.pending) DV_FAIL("dv_inst_config: a frame is in flight");
    T.max_cnt = max_dynamic_cnt; T.min_dist = min_dynamic_dist; T.use_det3d = use_det3d;
    std::vector<uint8_t> hw; circle_half_widths_i(min_dynamic_dist, hw);
    DV_CHECK(T.hw.ensure(DV_MAX_RADIUS + 1));
    DV_CHECK(hipMemcpy(T.hw.p, hw.data(), hw.size(), hipMemcpyHostToDevice));
    T.hw_radius = min_dynamic_dist;
    return 0;
}

int dv_inst_reset(dv_ctx* ctx) {
    if (!ctx) return -1;
    if (!ctx->inst) return 0;
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    DV_CHECK(hipStreamSynchronize(ctx->stream)); DV_CHECK(hipStreamSynchronize(ctx->inst->stream));
    for (auto& kv : ctx->inst->slots) kv.second.release();
    DV_CHECK(hipStreamSynchronize(ctx->inst->xstream));
    ctx->inst->disp_next = ctx->inst->xp_frame = ctx->inst->xp_inflight = false;
    ctx->inst->slots.clear(); ctx->inst->pending = false; ctx->inst->last_time = ctx->inst->cur_time = 0; ctx->inst->out_order.clear();
    return 0;
}

int dv_inst_track_enqueue(dv_ctx* ctx, double t, const dv_inst_det* dets, int n_dets, const dv_box3d* boxes3d, int n_boxes3d) {
    if (!ctx) return -1;
    HostScope hs(ctx, "h_inst_enqueue");
    if (!ctx->inst) DV_FAIL("dv_inst_track: call dv_inst_config first");
    if (n_dets < 0 || (n_dets > 0 && !dets) || n_boxes3d < 0 || (n_boxes3d > 0 && !boxes3d)) DV_FAIL("dv_inst_track: bad argument");
    if (!ctx->have_prev) DV_FAIL("dv_inst_track: enqueue the frame with dv_track_stereo_enqueue first (the object tracker works on its pyramids)");
    dv_inst_tracker& T = *ctx->inst;
    if (T.pending) DV_FAIL("dv_inst_track_enqueue: previous frame not collected");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    // the objects' stream: starts when this frame's pyramids exist (ev_pyr), its corner selection waits for the background tracker's (ev_bg_select) so
    // that the shared id counter is consumed in a fixed order; the background tracker's NEXT frame waits for T.done before it rebuilds the right pyramid
    hipStream_t s = T.stream;
    if (!ctx->ev_pyr || !ctx->ev_bg_select) DV_FAIL("dv_inst_track: internal events missing");
    DV_CHECK(hipStreamWaitEvent(s, ctx->ev_pyr, 0));
    if (T.xp_inflight) { DV_CHECK(hipStreamWaitEvent(s, T.ev_xdone, 0)); T.xp_inflight = false; }      // the previous frame's extra-point kernel read the masks this frame's uploads overwrite
    bool waited_bg = false;
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    const DvPyr& L = ctx->left[ctx->cur].pyr; const DvPyr& R = ctx->right.pyr;
    const bool stereo = ctx->cfg.stereo != 0;
    T.cur_time = t;
    // ---- FeatureTrack (system/main.cpp:198-212): every object starts the frame invisible; the detections (already associated to tracks) make them visible ----
    for (auto& kv : T.slots) { kv.second.visible = false; kv.second.has_box2d = false; kv.second.has_box3d = false; kv.second.points = nullptr; kv.second.n_points = 0; }
    size_t mask_bytes = 0;
    for (int i = 0; i < n_dets; ++i) {
        const dv_inst_det& d = dets[i];
        if (d.w <= 0 || d.h <= 0 || d.x < 0 || d.y < 0 || d.x + d.w > W || d.y + d.h > H) DV_FAIL("dv_inst_track: detection rectangle outside the image");
        if (!d.mask) DV_FAIL("dv_inst_track: detection without a mask");
        auto it = T.slots.find(d.track_id);
        if (it == T.slots.end()) {
            it = T.slots.emplace(d.track_id, Slot()).first; it->second.id = d.track_id;
            if (slot_init(ctx, it->second)) return -1;
        }
        Slot& S = it->second;
        if (S.visible) DV_FAIL("dv_inst_track: two detections with the same track id");
        S.visible = true; S.has_box2d = true; S.class_id = d.class_id; S.rx = d.x; S.ry = d.y; S.rw = d.w; S.rh = d.h;
        S.pts_copy.assign(d.points ? d.points : nullptr, d.points ? d.points + 3 * (size_t)std::max(d.n_points, 0) : nullptr);
        mask_bytes += (size_t)align_up(d.w, 16) * d.h;
    }
    // ---- InstsTrack (dynamic_tracker.cpp:348-493) ----
    for (auto& kv : T.slots) { if (!kv.second.visible) kv.second.lost_num++; else kv.second.lost_num = 0; }
    if (T.use_det3d) {       // BoxAssociate2Dto3D (:61-152): same class, IoU of the 2-D rectangles > 0.1, nearest centre wins, every 3-D box used once
        std::vector<char> taken(n_boxes3d, 0);
        for (auto& kv : T.slots) {
            Slot& S = kv.second;
            if (!S.visible) continue;
            const float ir[4] = { (float)S.rx, (float)S.ry, (float)S.rw, (float)S.rh };       // cv::Rect(min_pt, max_pt) of the detection
            double best = 1.7976931348623157e308; int bi = -1;
            for (int i = 0; i < n_boxes3d; ++i) {
                if (taken[i]) continue;
                const dv_box3d& b = boxes3d[i];
                if (b.class_id != S.class_id) continue;
                const int x0 = (int)lrintf(b.rect_min[0]), y0 = (int)lrintf(b.rect_min[1]), x1 = (int)lrintf(b.rect_max[0]), y1 = (int)lrintf(b.rect_max[1]);      // cv::Rect from Point2f: cvRound
                const float pr[4] = { (float)x0, (float)y0, (float)(x1 - x0), (float)(y1 - y0) };
                if (!(rect_iou(ir, pr) > 0.1f)) continue;
                const double dn = sqrt(b.center[0] * b.center[0] + b.center[1] * b.center[1] + b.center[2] * b.center[2]);
                if (dn < best) { best = dn; bi = i; }
            }
            if (bi >= 0) { taken[bi] = 1; S.has_box3d = true; S.box3d = boxes3d[bi]; }
        }
    }
    T.out_order.clear();
    if (n_dets == 0) { T.xp_frame = false; T.disp_next = false; }
    StageScope sc_all(ctx, "inst_track", s);
    if (n_dets > 0) {
        // masks: one pinned staging area, one H2D per object (sources are pageable caller memory)
        // (sized for two full frames of mask pixels from the start: the detections' rectangles GROW as objects come closer, and every growth step of these two buffers was a
        //  hipHostFree + hipHostMalloc resp. two stream synchronisations + hipFree + hipMalloc — 5 - 8 ms during which the estimator thread's HIP calls wait for the runtime's
        //  locks too; seen as one 6 ms frame somewhere in the first frames of the dynamic bench line's timed region in every second run)
        const size_t mask_floor = 2 * (size_t)align_up(W, 16) * H + 4096;
        if (T.pinned_in_bytes < mask_bytes) {
            if (T.pinned_in) (void)hipHostFree(T.pinned_in);
            T.pinned_in = nullptr; T.pinned_in_bytes = 0;
            const size_t want = std::max(mask_bytes * 2 + 4096, mask_floor);
            DV_CHECK(hipHostMalloc(&T.pinned_in, want, hipHostMallocDefault));
            T.pinned_in_bytes = want;
        }
        if (T.mask_all.bytes < mask_bytes) { DV_CHECK(hipStreamSynchronize(s)); DV_CHECK(hipStreamSynchronize(T.xstream)); DV_CHECK(T.mask_all.ensure(std::max(mask_bytes * 2 + 4096, mask_floor))); }
        // output area
        if (T.out_cap_slots < T.slots.size()) {
            const size_t cap = std::max<size_t>(8, T.slots.size() * 2);
            DV_CHECK(hipStreamSynchronize(s));
            DV_CHECK(T.out_buf.ensure(cap * (INST_CAP * sizeof(dv_feat) + 64)));
            if (T.out_pinned) (void)hipHostFree(T.out_pinned);
            T.out_pinned = nullptr;
            DV_CHECK(hipHostMalloc(&T.out_pinned, cap * (INST_CAP * sizeof(dv_feat) + 64), hipHostMallocDefault));
            T.out_cap_slots = cap;
        }
        const size_t slot_bytes = INST_CAP * sizeof(dv_feat) + 64;
        size_t moff = 0; int out_k = 0;
        const double dt = T.cur_time - T.last_time;
        // The objects are independent up to the shared id counter, and an LK launch is pure latency (one wave per point): the frame is processed in STAGES —
        // (A) the masks (one upload), ROI crops + paddings, pyramid levels; (B) temporal LK; (C) compaction, mask erosion (two passes), Shi-Tomasi tiles, corner
        // selection, ids (the objects draw on ONE counter: handed out in ascending object id by a one-workgroup launch behind the selection); (D) right-image LK;
        // (E) undistortion / velocities / rows.  Every stage is ONE launch for all objects (job tables filled on the host, one upload): 13 launches per frame
        // whatever the number of objects (they were 8 per object + 6).  Same arithmetic, same order of id assignment as the one-object-at-a-time form.
        std::vector<Slot*> act;
        for (auto& kv : T.slots) { Slot& S = kv.second; S.out_valid = false; if (S.lost_num > 0 || !S.visible) continue; act.push_back(&S); }      // ExecInst + is_curr_visible
        const int na = (int)act.size();
        if ((size_t)na * 2 > T.jobs_cap) {
            const size_t cap = std::max<size_t>(16, (size_t)na * 4);
            DV_CHECK(hipStreamSynchronize(s));
            DV_CHECK(T.jobs.ensure(cap * sizeof(DvLkJob)));
            if (T.jobs_pinned) (void)hipHostFree(T.jobs_pinned);
            T.jobs_pinned = nullptr;
            DV_CHECK(hipHostMalloc(&T.jobs_pinned, cap * sizeof(DvLkJob), hipHostMallocDefault));
            T.jobs_cap = cap;
        }
        DvLkJob* hj = (DvLkJob*)T.jobs_pinned; int n_temporal = 0;
        // ---- stage A: masks, then the ROI crops / re-paddings and the pyramid levels of ALL objects as one launch each (they were 1 + 2 + 3 launches per object, on
        // one stream: with four objects a chain of ~50 dependent launches per frame, ~0.8 ms of latency between the enqueue and the rows) ----
        {
            const size_t need = ((size_t)na * 2 * sizeof(RoiJob) + (size_t)na * DV_MAX_LEVELS * sizeof(DvPyrJob) + (size_t)na * sizeof(DvExtraJob) +
                                 (size_t)na * (sizeof(DvCompactJob) + sizeof(DvErodeJob) + sizeof(GfttTileArgs) + sizeof(GfttSelectArgs) + sizeof(DvFinalizeJob)) + 4096);
            if (need > T.arena_cap) {
                DV_CHECK(hipStreamSynchronize(s));
                const size_t cap = need * 2;
                DV_CHECK(T.arena.ensure(cap));
                if (T.arena_pinned) (void)hipHostFree(T.arena_pinned);
                T.arena_pinned = nullptr;
                DV_CHECK(hipHostMalloc(&T.arena_pinned, cap, hipHostMallocDefault));
                T.arena_cap = cap;
            }
        }
        RoiJob* h_roi = (RoiJob*)T.arena_pinned; int n_roi = 0, roi_W = 0, roi_H = 0;
        DvPyrJob* h_pyr = (DvPyrJob*)((uint8_t*)T.arena_pinned + (((size_t)na * 2 * sizeof(RoiJob) + 255) / 256) * 256);
        const size_t pyr_off = (uint8_t*)h_pyr - (uint8_t*)T.arena_pinned;
        int n_pyr[DV_MAX_LEVELS] = { 0 }, pyr_W[DV_MAX_LEVELS] = { 0 }, pyr_H[DV_MAX_LEVELS] = { 0 };      // per level l >= 1: jobs [l * na, l * na + n_pyr[l])
        // extra points of this frame from its disparity map: one job per visible object behind the pyramid jobs in the same upload
        const size_t xp_off = ((pyr_off + (size_t)na * DV_MAX_LEVELS * sizeof(DvPyrJob)) + 255) / 256 * 256;
        DvExtraJob* h_xp = (DvExtraJob*)((uint8_t*)T.arena_pinned + xp_off);
        const size_t xp_slot_bytes = 64 + (size_t)3 * DV_XP_CAP * sizeof(double);
        // the tables of stages C and E behind them
        auto up256 = [](size_t v) { return (v + 255) / 256 * 256; };
        const size_t cj_off = up256(xp_off + (size_t)na * sizeof(DvExtraJob)), ej_off = up256(cj_off + (size_t)na * sizeof(DvCompactJob)), gt_off = up256(ej_off + (size_t)na * sizeof(DvErodeJob)),
                     gs_off = up256(gt_off + (size_t)na * sizeof(GfttTileArgs)), fj_off = up256(gs_off + (size_t)na * sizeof(GfttSelectArgs)), arena_used = fj_off + (size_t)na * sizeof(DvFinalizeJob);
        DvCompactJob* h_cj = (DvCompactJob*)((uint8_t*)T.arena_pinned + cj_off); DvErodeJob* h_ej = (DvErodeJob*)((uint8_t*)T.arena_pinned + ej_off);
        GfttTileArgs* h_gt = (GfttTileArgs*)((uint8_t*)T.arena_pinned + gt_off); GfttSelectArgs* h_gs = (GfttSelectArgs*)((uint8_t*)T.arena_pinned + gs_off);
        DvFinalizeJob* h_fj = (DvFinalizeJob*)((uint8_t*)T.arena_pinned + fj_off);
        int ero_W = 0, ero_H = 0, nj = 0;
        const double dt_fin = dt;
        T.xp_frame = T.disp_next; T.disp_next = false;
        if (T.xp_frame && T.xp_cap_slots < (size_t)na) {
            DV_CHECK(hipStreamSynchronize(T.xstream));
            if (T.xp_pinned) (void)hipHostFree(T.xp_pinned);
            T.xp_pinned = nullptr; T.xp_cap_slots = 0;
            const size_t cap = std::max<size_t>(8, (size_t)na * 2);
            DV_CHECK(hipHostMalloc(&T.xp_pinned, cap * xp_slot_bytes, hipHostMallocDefault));
            T.xp_cap_slots = cap;
        }
        int n_xp = 0;
        for (Slot* Sp : act) {
            Slot& S = *Sp;
            const int w = S.rw, h = S.rh;
            const int mp = align_up(w, 16);
            uint8_t* hm = (uint8_t*)T.pinned_in + moff; moff += (size_t)mp * h;
            const dv_inst_det* det = nullptr; for (int i = 0; i < n_dets; ++i) if (dets[i].track_id == S.id) det = &dets[i];
            for (int y = 0; y < h; ++y) std::memcpy(hm + (size_t)y * mp, det->mask + (size_t)y * w, w);
            S.mask = (uint8_t*)T.mask_all.p + (hm - (uint8_t*)T.pinned_in);
            if (T.xp_frame) {          // ProcessExtraPoints visits the visible objects (ExecInst + is_curr_visible); slot k of the pinned output = k-th active object = its output index
                uint8_t* xo = (uint8_t*)T.xp_pinned + (size_t)n_xp * xp_slot_bytes;
                h_xp[n_xp++] = DvExtraJob{ (const uint8_t*)S.mask, mp, w, h, S.rx, S.ry, dv_extra_points_step(h, w), (double*)(xo + 64), (int*)xo };
            }
            // this frame's ROI, padded to the common size with the previous one (InstanceImagePadding)
            S.cur ^= 1;
            RoiPyr& B = S.roi[S.cur]; RoiPyr& Prev = S.roi[S.cur ^ 1];
            const int PW = S.has_prev ? std::max(w, S.pw) : w, PH = S.has_prev ? std::max(h, S.ph) : h;
            B.layout(PW, PH);
            {   // stage C / E tables of this object
                DvCompactJob cj{}; cj.tr = S.tr; cj.in_mask = nullptr; cj.mask_pitch = 0; cj.sort_by_cnt = 0; cj.n_cand = S.n_cand; cj.max_ord = S.max_ord;      // ReduceVector x4, ++track_cnt; without a previous ROI the object has no points (n_feat == 0)
                h_cj[nj] = cj;
                h_ej[nj] = DvErodeJob{ S.mask, (uint8_t*)S.tmp.p, (uint8_t*)S.ero.p, w, h, mp, mp, mp, 5 };      // ErodeMask 5x5 + discs of the tracked points + goodFeaturesToTrack on roi_gray (:418-446)
                ero_W = std::max(ero_W, w); ero_H = std::max(ero_H, h);
                GfttTileArgs a{};
                a.img = B.pyr.L[0].p; a.w = w; a.h = h; a.pitch = B.pyr.L[0].pitch;
                a.in_mask = (const uint8_t*)S.ero.p; a.mask_pitch = mp;
                a.disc_pts = S.tr.curr_pts; a.n_disc = S.tr.n_tracked; a.radius = T.min_dist; a.hw = (const uint8_t*)T.hw.p;
                a.n_feat = S.tr.n_feat; a.max_cnt = T.max_cnt; a.min_new = 1;
                a.cand = (DvCand*)S.cand.p; a.cand_cap = T.cand_cap; a.n_cand = S.n_cand; a.max_ord = S.max_ord;
                h_gt[nj] = a;
                GfttSelectArgs sa{};
                sa.cand = (const DvCand*)S.cand.p; sa.n_cand = S.n_cand; sa.cand_cap = T.cand_cap; sa.max_ord = S.max_ord;
                sa.w = w; sa.h = h; sa.quality = 0.01; sa.min_dist = (double)T.min_dist;
                sa.max_n_host = 0; sa.n_feat = S.tr.n_feat; sa.max_cnt = T.max_cnt; sa.min_new = 1;
                sa.out_xy = nullptr; sa.n_out = nullptr; sa.tr = S.tr; sa.has_tr = 2; sa.err_flag = T.err_flag; sa.id_slot = S.id_slot;
                h_gs[nj] = sa;
                dv_feat* od = (dv_feat*)((uint8_t*)T.out_buf.p + (size_t)nj * slot_bytes);
                DvFinalizeJob f{}; f.tr = S.tr; f.cam0 = ctx->cfg.cam0; f.cam1 = ctx->cfg.cam1; f.stereo = stereo ? 1 : 0; f.use_off = 1; f.dt = dt_fin; f.out = od;
                f.n_out = (int*)((uint8_t*)od + INST_CAP * sizeof(dv_feat)); f.err_in = nullptr; f.err_out = nullptr; f.off_x = (float)(double)S.rx; f.off_y = (float)(double)S.ry;
                h_fj[nj] = f;
                ++nj;
            }
            h_roi[n_roi++] = RoiJob{ L.L[0].p, B.pyr.L[0].p, L.L[0].pitch, S.rx, S.ry, w, h, B.pyr.L[0].pitch, PW, PH };
            roi_W = std::max(roi_W, PW); roi_H = std::max(roi_H, PH);
            if (S.has_prev) {
                S.padA.layout(PW, PH);
                h_roi[n_roi++] = RoiJob{ Prev.pyr.L[0].p, S.padA.pyr.L[0].p, Prev.pyr.L[0].pitch, 0, 0, S.pw, S.ph, S.padA.pyr.L[0].pitch, PW, PH };
                for (int l = 1; l < B.pyr.levels; ++l) {
                    h_pyr[(size_t)l * na + n_pyr[l]++] = DvPyrJob{ S.padA.pyr.L[l - 1].p, B.pyr.L[l - 1].p, S.padA.pyr.L[l].p, B.pyr.L[l].p, B.pyr.L[l - 1].w, B.pyr.L[l - 1].h, B.pyr.L[l - 1].pitch,
                                                                     B.pyr.L[l].w, B.pyr.L[l].h, B.pyr.L[l].pitch };
                    pyr_W[l] = std::max(pyr_W[l], B.pyr.L[l].w); pyr_H[l] = std::max(pyr_H[l], B.pyr.L[l].h);
                }
                // InstFeat::TrackLeft: FeatureTrackByLK(prev padded, cur padded, last_points) without a mask (dynamic_tracker.cpp:409)
                DvLkJob j{}; j.A = S.padA.pyr; j.B = B.pyr; j.pts_a = S.tr.last_pts; j.n_dev = S.tr.n_feat; j.pts_b = S.tr.lk_pts; j.status = S.tr.lk_status;
                hj[n_temporal++] = j;
            }
        }
        if (na > 0) {
            DV_CHECK(dv_copy_async(T.mask_all.p, T.pinned_in, moff, s));
            DV_CHECK(dv_copy_async(T.arena.p, T.arena_pinned, arena_used, s));
            if (T.xp_frame && n_xp > 0) {
                // the reference starts a thread for this (dynamic_tracker.cpp:378): a side stream behind the mask + table uploads; the objects' tracking goes on meanwhile
                DV_CHECK(hipEventRecord(T.ev_xin, s));
                DV_CHECK(hipStreamWaitEvent(T.xstream, T.ev_xin, 0));
                const float* dmap = T.disp_user; int dpitch = T.disp_stride / 4;
                if (T.disp_mem != DV_MEM_DEVICE) {
                    DV_CHECK(T.disp_buf.ensure((size_t)W * H * 4));
                    DV_CHECK(hipMemcpy2DAsync(T.disp_buf.p, (size_t)W * 4, T.disp_user, (size_t)T.disp_stride, (size_t)W * 4, H, hipMemcpyHostToDevice, T.xstream));
                    dmap = (const float*)T.disp_buf.p; dpitch = W;
                }
                if (T.xp_pool.bytes < dv_extra_points_scratch_bytes(n_xp)) { DV_CHECK(hipStreamSynchronize(T.xstream)); DV_CHECK(T.xp_pool.ensure(dv_extra_points_scratch_bytes(std::max(8, 2 * n_xp)))); }
                DvExtraArgs xa{ dmap, dpitch, W, H, (float)ctx->cfg.cam0.fx, (float)ctx->cfg.cam0.fy, (float)ctx->cfg.cam0.cx, (float)ctx->cfg.cam0.cy, (float)T.disp_baseline, T.err_flag, 0, (uint8_t*)T.xp_pool.p };
                StageScope scx(ctx, "inst_extra_points", T.xstream);
                if (dv_launch_extra_points((const DvExtraJob*)((const uint8_t*)T.arena.p + xp_off), n_xp, xa, T.xstream)) DV_FAIL("extra_points: cannot set dynamic LDS size");
                T.xp_inflight = true;
            }
            { dim3 grid((roi_W + 255) / 256, roi_H, n_roi); hipLaunchKernelGGL(roi_pad_multi_kernel, grid, dim3(256), 0, s, (const RoiJob*)T.arena.p); }
            const DvPyrJob* d_pyr = (const DvPyrJob*)((const uint8_t*)T.arena.p + pyr_off);
            for (int l = 1; l < DV_MAX_LEVELS; ++l) dv_launch_pyr_down_multi(d_pyr + (size_t)l * na, n_pyr[l], pyr_W[l], pyr_H[l], s);
        }
        // the right-image jobs are known up front as well: TrackRightByPad (instance_feature.cpp:251-275) moves the points into full-image coordinates
        for (int k = 0; k < na && stereo; ++k) {
            Slot& S = *act[k];
            DvLkJob j{}; j.A = L; j.B = R; j.pts_a = S.tr.curr_pts; j.n_dev = S.tr.n_feat; j.pts_b = S.tr.right_pts; j.status = S.tr.right_status;
            j.add_x = (float)S.rx; j.add_y = (float)S.ry; j.use_add = 1; j.key = S.id;
            hj[n_temporal + k] = j;
        }
        if (na > 0) DV_CHECK(dv_copy_async(T.jobs.p, hj, (size_t)(n_temporal + (stereo ? na : 0)) * sizeof(DvLkJob), s));
        // ---- stage B ----
        dv_launch_lk_track_multi((const DvLkJob*)T.jobs.p, n_temporal, T.max_cnt, ctx->cfg.flow_back, 0.5f, s);
        // ---- stage C ----
        if (na > 0) {
            const uint8_t* ap = (const uint8_t*)T.arena.p;
            dv_launch_compact_multi((const DvCompactJob*)(ap + cj_off), na, s);
            dv_launch_erode_multi((const DvErodeJob*)(ap + ej_off), na, ero_W, ero_H, s);
            dv_launch_gftt_tile_multi((const GfttTileArgs*)(ap + gt_off), na, ero_W, ero_H, s);
            if (!waited_bg) { DV_CHECK(hipStreamWaitEvent(s, ctx->ev_bg_select, 0)); waited_bg = true; }      // ids: background first, then the objects in ascending id
            if (dv_launch_gftt_select_multi((const GfttSelectArgs*)(ap + gs_off), na, s)) DV_FAIL("gftt_select: cannot set dynamic LDS size");
            dv_launch_gftt_assign_ids((const GfttSelectArgs*)(ap + gs_off), na, s);
        }
        // ---- stage D ----
        if (stereo) dv_launch_lk_track_multi((const DvLkJob*)T.jobs.p + n_temporal, na, T.max_cnt, ctx->cfg.flow_back, 0.5f, s);
        if (stereo && T.keys_next && na > 0) {          // cfg::dataset == kViode: the segmentation-key test of TrackRightByPad (instance_feature.cpp:263-268)
            const uint32_t* kimg = T.keys_user; int kpitch = T.keys_stride / 4;
            if (T.keys_mem != DV_MEM_DEVICE && T.keys_mem != DV_MEM_PINNED) {
                DV_CHECK(T.keys_buf.ensure((size_t)W * H * 4));
                DV_CHECK(hipMemcpy2DAsync(T.keys_buf.p, (size_t)W * 4, T.keys_user, (size_t)T.keys_stride, (size_t)W * 4, H, hipMemcpyHostToDevice, s));
                kimg = (const uint32_t*)T.keys_buf.p; kpitch = W;
            }
            dv_launch_right_key_check((const DvLkJob*)T.jobs.p + n_temporal, na, kimg, kpitch, W, H, s);
        }
        // ---- stage E: UndistortedPointsWithAddOffset + PtsVelocity + RightUndistortedPts + RightPtsVelocity + PostProcess -> rows ----
        if (na > 0) dv_launch_finalize_multi((const DvFinalizeJob*)((const uint8_t*)T.arena.p + fj_off), na, T.max_cnt, s);
        for (Slot* Sp : act) {
            Slot& S = *Sp;
            S.has_prev = true; S.pw = S.rw; S.ph = S.rh;            // PostProcess: prev_roi_gray = roi_gray (the top-left w x h of roi[cur] level 0)
            S.out_valid = true; S.out_index = out_k++;
            T.out_order.push_back(S.id);
        }
        if (out_k > 0) DV_CHECK(dv_copy_async(T.out_pinned, T.out_buf.p, (size_t)out_k * slot_bytes, s));
    }
    // ---- ManageInstances (:499-514): objects unseen for more than one frame are dropped ----
    for (auto it = T.slots.begin(); it != T.slots.end();) {
        Slot& S = it->second;
        if (S.lost_num == 0 && !S.has_box2d) S.lost_num++;
        bool erase = false;
        if (S.lost_num > 0) { S.lost_num++; if (S.lost_num > 3) erase = true; }
        if (erase) { DV_CHECK(hipStreamSynchronize(s)); S.release(); it = T.slots.erase(it); } else ++it;
    }
    DV_CHECK(hipGetLastError());
    T.keys_next = false;          // (the key image belongs to ONE frame, like the disparity map)
    if (T.xp_inflight) DV_CHECK(hipEventRecord(T.ev_xdone, T.xstream));
    DV_CHECK(hipEventRecord(T.done, s));
    T.last_time = T.cur_time; T.pending = true; T.frame_enqueued = true;
    return 0;
}

// VIODE: the keys of SemanticImage::seg1 of the frame the NEXT dv_inst_track_enqueue processes (include/dvins.h)
int dv_inst_set_right_keys(dv_ctx* ctx, const uint32_t* key_image, int stride_bytes, int mem) {
    if (!ctx) return -1;
    if (!ctx->inst) DV_FAIL("dv_inst_set_right_keys: call dv_inst_config first");
    dv_inst_tracker& T = *ctx->inst;
    if (!key_image) { T.keys_next = false; return 0; }
    if (stride_bytes == 0) stride_bytes = 4 * ctx->cfg.width;
    if (stride_bytes < 4 * ctx->cfg.width || (stride_bytes & 3)) DV_FAIL("dv_inst_set_right_keys: bad stride");
    T.keys_user = key_image; T.keys_stride = stride_bytes; T.keys_mem = mem; T.keys_next = true;
    return 0;
}

// SemanticImage::disp of the frame the NEXT dv_inst_track_enqueue processes (include/dvins.h)
int dv_inst_set_disparity(dv_ctx* ctx, const float* disp, int stride_bytes, int mem, double baseline) {
    if (!ctx) return -1;
    if (!ctx->inst) DV_FAIL("dv_inst_set_disparity: call dv_inst_config first");
    dv_inst_tracker& T = *ctx->inst;
    if (!disp) { T.disp_next = false; return 0; }
    if (stride_bytes == 0) stride_bytes = 4 * ctx->cfg.width;
    if (stride_bytes < 4 * ctx->cfg.width || (stride_bytes & 3) || !(baseline > 0)) DV_FAIL("dv_inst_set_disparity: bad stride / baseline");
    T.disp_user = disp; T.disp_stride = stride_bytes; T.disp_mem = mem; T.disp_baseline = baseline; T.disp_next = true;
    return 0;
}

// one object through DetectExtraPoints (+ ProcessExtraPoints' point-cloud half): operator form for the parity tests
int dv_extra_points(dv_ctx* ctx, const uint8_t* mask, int x, int y, int w, int h, const float* disp, int stride_bytes, int mem, double baseline, int stage,
                    double* out_xyz, int cap_out, int* n_out) {
    if (!ctx) return -1;
    if (!mask || !disp || !out_xyz || !n_out || w <= 0 || h <= 0 || cap_out < 0 || (stage != 0 && stage != 1)) DV_FAIL("dv_extra_points: bad argument");
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    if (stride_bytes == 0) stride_bytes = 4 * W;
    if (stride_bytes < 4 * W || (stride_bytes & 3)) DV_FAIL("dv_extra_points: bad stride");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    hipStream_t s = ctx->stream;
    const int mp = align_up(w, 16);
    DV_CHECK(ctx->s0.ensure((size_t)mp * h));
    DV_CHECK(hipMemcpy2DAsync(ctx->s0.p, mp, mask, w, w, h, hipMemcpyHostToDevice, s));
    const float* dmap = disp; int dpitch = stride_bytes / 4;
    if (mem != DV_MEM_DEVICE) {
        DV_CHECK(ctx->s1.ensure((size_t)W * H * 4));
        DV_CHECK(hipMemcpy2DAsync(ctx->s1.p, (size_t)W * 4, disp, (size_t)stride_bytes, (size_t)W * 4, H, hipMemcpyHostToDevice, s));
        dmap = (const float*)ctx->s1.p; dpitch = W;
    }
    const size_t out_bytes = 64 + (size_t)3 * DV_XP_CAP * sizeof(double);
    DV_CHECK(ctx->s2.ensure(out_bytes + sizeof(DvExtraJob) + 256));
    uint8_t* ob = (uint8_t*)ctx->s2.p;
    DvExtraJob job{ (const uint8_t*)ctx->s0.p, mp, w, h, x, y, dv_extra_points_step(h, w), (double*)(ob + 64), (int*)ob };
    DvExtraJob* jd = (DvExtraJob*)(ob + out_bytes + 128);
    DV_CHECK(hipMemcpyAsync(jd, &job, sizeof(job), hipMemcpyHostToDevice, s));
    DV_CHECK(hipMemsetAsync(ctx->err_flag, 0, 4, s));
    DV_CHECK(ctx->s3.ensure(dv_extra_points_scratch_bytes(1)));
    DvExtraArgs xa{ dmap, dpitch, W, H, (float)ctx->cfg.cam0.fx, (float)ctx->cfg.cam0.fy, (float)ctx->cfg.cam0.cx, (float)ctx->cfg.cam0.cy, (float)baseline, ctx->err_flag, stage, (uint8_t*)ctx->s3.p };
    if (dv_launch_extra_points(jd, 1, xa, s)) DV_FAIL("extra_points: cannot set dynamic LDS size");
    DV_CHECK(hipGetLastError());
    int n = 0, ef = 0;
    DV_CHECK(hipMemcpyAsync(&n, ob, 4, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipMemcpyAsync(&ef, ctx->err_flag, 4, hipMemcpyDeviceToHost, s));
    DV_CHECK(hipStreamSynchronize(s));
    if (ef) { DV_CHECK(hipMemsetAsync(ctx->err_flag, 0, 4, s)); DV_FAIL("dv_extra_points: more than DV_XP_CAP sampled points (8) or clustering not converged (16) (device error flags=" + std::to_string(ef) + ")"); }
    if (n > cap_out) DV_FAIL("dv_extra_points: output buffer too small");
    if (n > 0) DV_CHECK(hipMemcpy(out_xyz, ob + 64, (size_t)n * 24, hipMemcpyDeviceToHost));
    *n_out = n;
    return 0;
}

// InstsFeatManager::Output() (:521-577): one FeatureInstance per visible object
int dv_inst_track_collect(dv_ctx* ctx, dv_inst_obs* insts, int cap_insts, int* n_insts, dv_feat* feats, int cap_feats, int* n_feats, double* points, int cap_points, int* n_points) {
    if (!ctx) return -1;
    HostScope hs(ctx, "h_inst_collect");
    if (!ctx->inst) DV_FAIL("dv_inst_track: call dv_inst_config first");
    dv_inst_tracker& T = *ctx->inst;
    if (!T.pending) DV_FAIL("dv_inst_track_collect: nothing enqueued");
    if (!n_insts || !n_feats || !n_points) DV_FAIL("dv_inst_track_collect: null counter");
    { HostScope hw(ctx, "h_inst_wait"); DV_CHECK(hipEventSynchronize(T.done)); if (T.xp_frame && T.xp_inflight) DV_CHECK(hipEventSynchronize(T.ev_xdone)); }      // the object tracker's launch chain of this frame (+ the extra-point side stream)
    T.pending = false;
    if (ctx->timing) { DV_CHECK(hipStreamSynchronize(T.stream)); dv_harvest_timers(ctx, T.stream); DV_CHECK(hipStreamSynchronize(T.xstream)); dv_harvest_timers(ctx, T.xstream); }
    const size_t xp_slot_bytes = 64 + (size_t)3 * DV_XP_CAP * sizeof(double);
    const size_t slot_bytes = INST_CAP * sizeof(dv_feat) + 64;
    int ki = 0, kf = 0, kp = 0;
    for (unsigned id : T.out_order) {
        auto it = T.slots.find(id);
        if (it == T.slots.end()) continue;
        Slot& S = it->second;
        if (!S.out_valid) continue;
        const dv_feat* rows = (const dv_feat*)((const uint8_t*)T.out_pinned + (size_t)S.out_index * slot_bytes);
        const int n = *(const int*)((const uint8_t*)rows + INST_CAP * sizeof(dv_feat));
        // extra points: computed on the device from the frame's disparity map (slot = output index), or the caller's (pass-through)
        const uint8_t* xs = T.xp_frame ? (const uint8_t*)T.xp_pinned + (size_t)S.out_index * xp_slot_bytes : nullptr;
        const int np = xs ? *(const int*)xs : (int)(S.pts_copy.size() / 3);
        const double* psrc = xs ? (const double*)(xs + 64) : S.pts_copy.data();
        if (ki >= cap_insts || kf + n > cap_feats || kp + np > cap_points) DV_FAIL("dv_inst_track_collect: output buffers too small");
        dv_inst_obs& o = insts[ki++];
        std::memset(&o, 0, sizeof(o));
        o.id = S.id; o.has_box3d = S.has_box3d ? 1 : 0; o.first_feat = kf; o.n_feats = n; o.first_point = kp; o.n_points = np;
        o.rect[0] = (float)S.rx; o.rect[1] = (float)S.ry; o.rect[2] = (float)S.rw; o.rect[3] = (float)S.rh;
        if (S.has_box3d) o.box3d = S.box3d;
        if (n > 0) std::memcpy(feats + kf, rows, (size_t)n * sizeof(dv_feat));
        if (np > 0) std::memcpy(points + 3 * (size_t)kp, psrc, 24 * (size_t)np);
        kf += n; kp += np;
    }
    *n_insts = ki; *n_feats = kf; *n_points = kp;
    if (*ctx->err_pinned) { /* reported by the next dv_track_stereo_collect */ }
    return 0;
}

}  // extern "C"
