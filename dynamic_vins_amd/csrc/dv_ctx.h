// dv_ctx.h — private context definition shared by the front-end and back-end ABI implementations.
#pragma once
#include "dv_internal.h"
#include "be_types.h"
#include "be_kernels.h"
#include <cmath>
#include <cstring>
#include <chrono>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

struct PyrSet {
    DvPyr pyr{}; DevBuf buf; int w = 0, h = 0, ml = -1; bool all = false;
    // all_levels: every one of the max_level + 1 levels (cv::cuda::SparsePyrLKOpticalFlow builds them unconditionally); default: buildOpticalFlowPyramid's stop rule
    hipError_t alloc(int w_, int h_, int max_level, bool all_levels = false) {
        if (w == w_ && h == h_ && ml == max_level && all == all_levels && pyr.levels > 0) return hipSuccess;
        w = w_; h = h_; ml = max_level; all = all_levels;
        int lw[DV_MAX_LEVELS], lh[DV_MAX_LEVELS], n = 0;
        int cw = w, ch = h;
        for (int l = 0; l <= max_level && l < DV_MAX_LEVELS; ++l) {
            lw[n] = cw; lh[n] = ch; ++n;
            int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
            if (!all_levels && (nw <= DV_LK_WIN || nh <= DV_LK_WIN)) break;      // buildOpticalFlowPyramid stop rule
            if (nw < 1 || nh < 1) break;
            cw = nw; ch = nh;
        }
        size_t total = 0, off[DV_MAX_LEVELS];
        const int A = DV_PYR_APRON;                                 // every level carries a reflect-101 apron: all LK tiles take the aligned dword path (lk.hip)
        for (int l = 0; l < n; ++l) { off[l] = total; total += (size_t)align_up(lw[l] + 2 * A, 16) * (lh[l] + 2 * A); total = (total + 255) / 256 * 256; }
        total += 512;                                             // slack for aligned over-reads
        hipError_t e = buf.ensure(total);
        if (e != hipSuccess) return e;
        for (int l = 0; l < n; ++l) { const int pitch = align_up(lw[l] + 2 * A, 16); pyr.L[l] = DvLevel{ (uint8_t*)buf.p + off[l] + (size_t)A * pitch + A, lw[l], lh[l], pitch, A }; }
        for (int l = n; l < DV_MAX_LEVELS; ++l) pyr.L[l] = DvLevel{ nullptr, 0, 0, 0, 0 };
        pyr.levels = n;
        return hipSuccess;
    }
};

struct StageTimer {       // pool of HIP event pairs: a stage / kernel may be recorded many times between two harvests
    std::string name; hipStream_t stream = nullptr; std::vector<std::pair<hipEvent_t, hipEvent_t>> pool; size_t used = 0; double total_ms = 0; long long count = 0;
};

// device-side workspace of the bundle-adjustment solver (allocated on first use)
// marginalization index tables (int32): prior_map[BE_MAX_PRIOR] | imu_map[32] | dim_slot[256] | dim_comp[256] | lm_sel[BE_MAX_LM]
#define BE_MT_PRIOR 0
#define BE_MT_IMU (BE_MAX_PRIOR)
#define BE_MT_SLOT (BE_MAX_PRIOR + 32)
#define BE_MT_COMP (BE_MAX_PRIOR + 32 + 256)
#define BE_MT_SEL (BE_MAX_PRIOR + 32 + 512)
#define BE_MARG_TAB_INTS (BE_MAX_PRIOR + 32 + 512 + BE_MAX_LM)

struct MargPlan {
    int mode = -1, D = 0, m = 0, n = 0, nimu = 0, nsel = 0;
    bool empty = false;                                       // nothing to drop: the prior becomes invalid
    int pose_dim[BE_NF], sb_dim[BE_NF], ex_dim[2], td_dim;
    int32_t tab[BE_MARG_TAB_INTS];                            // device image of the index tables
};

struct BePending {        // a solve that has been enqueued and not yet collected (be_solve_fused_begin / _end)
    bool want_raw_pose = false;
    bool deferred = false; int first_slots = 0;      // member of a dv_batch: uploaded, the slots wait for dv_batch_enqueue
    bool rej_on = false; BeRejectArgs rej{};      // this frame's device-side outlier test
    hipEvent_t ev_state_ext = nullptr;   // member of a dv_batch round with shared tail launches: the batch's event replaces BeWork::ev_state for this frame
    bool fuse_accept_gauge = false;      // estimator path outside a dv_batch: the last slot's accept decision rides in the gauge kernel (be_accept_gauge_kernel)
    bool active = false, trivial = false, do_marg = false, fused_present = false, marg_in_flight = false, marg_check_due = false; int scal_slot = 0, check_slot = 0; size_t state_bytes = 0; int nxt = 0; MargPlan pl;
    BeExt xt{}; bool copy_ex_td = false;      // free extrinsic / td blocks of this solve (be_kernels.h); copy_ex_td: hand the solved blocks back through dv_ba_problem::ex_pose / td
    BeEvalArgs ea; BeSolveArgs sa; int max_iters = 0, nframes = 0, use_imu = 0, nlm = 0; double g_norm = 0, gauge_R0[9], gauge_ypr0[3], gauge_P0[3];
    std::chrono::steady_clock::time_point t_begin, t_up, t_enq;
};

struct BeWork {
    std::unique_ptr<BePending> pend = std::make_unique<BePending>();
    bool ready = false;
    DevBuf block;          // one allocation, carved below
    BeCtl* ctl = nullptr; BeState* x = nullptr; BeState* cand = nullptr;
    BeFactor* fac = nullptr; BeLm* lm = nullptr; BeImu* imu = nullptr; BePriorHdr* prior = nullptr; double* priorA = nullptr; double* priorb = nullptr;
    double* packets[2] = { nullptr, nullptr }; double* imu_out[2] = { nullptr, nullptr }; double* prior_out[2] = { nullptr, nullptr }; double* cand_cost = nullptr; int32_t* lm_obs = nullptr;
    double* Hd[2] = { nullptr, nullptr }; double* Sc[2] = { nullptr, nullptr }; double* gvec[2] = { nullptr, nullptr };      // two linearisation sets (BeCtl::cur)
    double* scale_p = nullptr; double* diag_p = nullptr; double* grad_p = nullptr; double* gn_p = nullptr;
    double* scale_l = nullptr; double* diag_l = nullptr; double* grad_l = nullptr; double* gn_l = nullptr;
    int32_t* prior_col = nullptr; int32_t* col_kind = nullptr; int32_t* col_frame = nullptr; int32_t* col_comp = nullptr;
    int fac_cap = 0;
    DevBuf marg_buf;       // per-landmark slabs of the marginalization (sized on demand)
    DevBuf xpk_buf;        // ext packets of both linearisation sets (free extrinsic / td blocks, be_ext.hip): allocated by the first solve that frees one
    void* pinned = nullptr; size_t pinned_bytes = 0;      // host staging: mirror of the device's upload region + download area
    size_t up_ctl = 0, up_x = 0, up_imu = 0, up_prior = 0, up_idx = 0, up_mt = 0, up_lm = 0, up_fac = 0, dl_off = 0;
    double* priorA_buf[2] = { nullptr, nullptr }; double* priorb_buf[2] = { nullptr, nullptr }; int prior_cur = 0;      // double-buffered prior (A', b')
    bool prior_resident = false;      // true: buffer prior_cur holds the estimator's current prior (written by the fused marginalization)
    hipEvent_t ev_state = nullptr;    // recorded behind the download of the solved states (the marginalization runs on past it)
    hipStream_t c0_stream = nullptr; hipEvent_t ev_margA = nullptr, ev_c0 = nullptr; bool c0_side = false, c0_pending = false;     // dv_debug_set "c0_side": the prior's constant c0 on a side stream (be_api.hip marg_enqueue); measured SLOWER (969-977 against 1008 frames/s: the two cross-stream event waits cost more than the 33 us they hide), so off
    double* prior_c0 = nullptr;       // [2] the prior's constant c0 per buffer, device resident
    int32_t* marg_tab = nullptr; double* marg_scal = nullptr;      // marginalization index tables (inside the upload region) and its 4 result scalars
    long long marg_checked = 0; double marg_last[4] = { 0, 0, 0, 0 };      // marginalizations whose health scalars came back, and the last set (c0, smallest pivot of A_mm, clamp flag, rank)
    long long marg_clamped = 0;       // marginalizations in which a pivot of A_mm was <= 1e-8 and was skipped (pseudo-inverse)
    bool debug_short_first_pass = false, ldl_generic = false, debug_wait_tail = false; int debug_batch_single = 0; bool debug_hash_log = false, debug_hash_light = false;
    // dv_debug_set "hash_log": hashes of what the DEVICE holds, per window solve: [counter, uploaded block as it arrived, prior A / b / c0 on the device at the start, x after the
    // round (raw solution), candidate buffer (gauge-fixed copy), control block]; written by be_dbg_hash_kernel into pinned slots, collected at the end of the solve
    std::vector<unsigned long long> dbg_dev_log; unsigned long long* dbg_pinned = nullptr; unsigned long long dbg_solve_no = 0;
    // "hash_log", finer: after EVERY launch of a round (slot it, launch kind 0 head-eval | 1 head-reduce | 2 solve | 3 candidate-eval | 4 candidate-reduce) hashes of everything that
    // launch may write: DBG_RANGES values at dbg_slots[(it * 5 + kind) * DBG_RANGES + r] (pinned); appended to dbg_slot_log per solve as [counter, slots * 5 * DBG_RANGES values]
    static constexpr int DBG_RANGES = 16, DBG_SLOTS = 16;
    unsigned long long* dbg_slots = nullptr; std::vector<unsigned long long> dbg_slot_log;      // dv_debug_set
    bool gpu_reject = true; uint8_t* rej_pinned = nullptr;      // dv_debug_set "gpu_reject": the outlier test of the frame on the device, flags written to pinned memory before ev_state
    std::vector<const double*> sqrt_hint;                  // optional cached IMU sqrt-information per factor (set by the estimator around a solve)
};

bool be_imu_sqrt_info(const double* cov15x15, double* U15x15);      // U^T U = cov^-1, false if singular

// Estimator-internal fused entry: window solve, yaw-gauge fix and marginalization enqueued back to back on the BA
// stream (one upload, one download, one sync); the new prior's A', b' stay in HBM for the next solve.
struct BeFused {
    int marg_mode = -1;                       // -1 none, 0 kMarginOld, 1 kMarginSecondNew
    double R0[9], ypr0[3], P0[3];             // Rs[0], R2ypr(Rs[0]), Ps[0] before the solve (gauge reference)
    dv_ba_prior new_prior;                    // out (marg_mode >= 0): header of the new prior (valid may be 0)
    double diag[4] = { 0, 0, 0, 0 };          // out: c0, smallest pivot, failure flag, rank
    bool want_raw_pose = false;               // in
    bool want_reject = false; double rej_ric[2][9], rej_tic[2][3], rej_focal = 0; const uint8_t* rej_flags = nullptr;      // in: OutliersRejection on the device (be_reject_kernel) with the extrinsics the host will hold after Double2vector; out: one flag per landmark of the problem (nullptr: not run)
    double raw_pose[77];                      // out: the solver's pose blocks BEFORE the yaw-gauge fix (what body.para_pose holds after ceres::Solve)
};
int be_solve_fused(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary, BeFused* fused);
void be_batch_detach(dv_ctx* ctx);          // dv_destroy: leave the dv_batch this ctx is a member of
struct DvFrontBatch;                        // the front-end half of a dv_batch (dvins_api.hip): stream, event, job tables of the shared tracking launches
DvFrontBatch*& be_batch_front(struct dv_batch* B);
int be_batch_index(struct dv_batch* B);
const std::vector<dv_ctx*>& be_batch_members(struct dv_batch* B);
void dv_front_batch_release(DvFrontBatch* F);
void dv_front_batch_sync(DvFrontBatch* F);
void* be_staging_factors(dv_ctx* ctx, int* cap);      // where the next solve's upload reads its factor table (pinned); nullptr if the workspace cannot be set up
int be_solve_fused_begin(dv_ctx* ctx, dv_ba_problem* P, BeFused* fused);       // upload + enqueue everything, returns immediately
int be_solve_fused_end(dv_ctx* ctx, dv_ba_problem* P, dv_ba_summary* summary, BeFused* fused);      // sync + collect
struct dv_obj_problem;
struct ObjPending {       // an object solve in flight (be_obj_solve_begin / _end): pinned staging for upload + download, its event
    bool active = false; void* pinned = nullptr; size_t pinned_bytes = 0, up_bytes = 0; hipEvent_t ev = nullptr; hipStream_t stream = nullptr; int V = 0, nblk = 0, n_obj = 0;
    void release() { if (pinned) (void)hipHostFree(pinned); pinned = nullptr; pinned_bytes = 0; if (ev) (void)hipEventDestroy(ev); ev = nullptr; active = false; }
};
int be_obj_solve_begin(dv_ctx* ctx, dv_obj_problem* P, hipStream_t s, DevBuf& scratch, ObjPending& pend);
int be_obj_solve_end(dv_ctx* ctx, dv_obj_problem* P, dv_ba_summary* summary, ObjPending& pend);
int be_obj_solve_prepare(dv_ctx* ctx, DevBuf& scratch, ObjPending& pend);      // the buffers, event and kernel attribute be_obj_solve_begin would create on its first call
int be_prepare(dv_ctx* ctx, bool dynamic);      // dv_est_create: everything the first window solve / marginalization / object solve would allocate or create lazily (a 3 ms frame otherwise)
struct dv_estimator;
void dv_est_destroy_internal(dv_estimator* e);
struct dv_inst_tracker;
void dv_inst_destroy_internal(dv_inst_tracker* t);
int dv_inst_wait_before_next_frame(dv_ctx* ctx);

// One window sharded by landmark over several GPUs (be_shard.hip): the transport of the exchange vectors
struct DvDist {
    int transport = 0;            // 0 none, 1 RCCL all-gather on the BA stream, 2 host call-back (staged through pinned memory), 3 one-shot peer writes (be_shard.hip)
    int rank = 0, world = 1;
    void* comm = nullptr;         // ncclComm_t
    dv_allgather_fn fn = nullptr; void* user = nullptr;
    DevBuf xsend, xrecv, qf; void* h_send = nullptr; void* h_recv = nullptr;      // qf: the summed form coefficients, one image per linearisation set (2 x BE_QF_LEN doubles)
    long long exchanges = 0;
    // transport 3: the window this rank exposes to its peers (hipIpc): [2 parities][world][slot] doubles, then [2][world] sequence flags, then a time-out flag;
    // peer_win[r] = rank r's window as mapped into this process (own window for r == rank)
    void* win = nullptr; size_t win_slot = 0; void* peer_win[64] = { nullptr }; unsigned long long seq = 0;
    // a wait kernel that gave up on a peer writes the exchange's sequence number here (pinned host memory, sticky): every collect of a solve reads it and fails instead of
    // consuming the gathered buffer; later wait kernels see it and do not wait again (one time-out per dead peer, not one per exchange)
    unsigned long long* peer_dead = nullptr; long long peer_timeout_ticks = 200000000ll;      // wall_clock64 ticks (100 MHz): 2 s; dv_debug_set "peer_timeout_ms"
};
int be_dist_check(dv_ctx* ctx);      // 0 = every exchange so far was complete; -1 + error: a peer never delivered (transport peer)
int be_exchange(dv_ctx* ctx, size_t count, hipStream_t s);      // all-gather `count` doubles of dist.xsend into dist.xrecv (rank-major), ordered on s
int be_dist_buffers(dv_ctx* ctx);
void be_dist_release(dv_ctx* ctx);

struct dv_ctx {
    dv_config cfg{};
    std::string err;
    hipStream_t stream = nullptr;
    PyrSet left[2], right; int cur = 0; bool have_prev = false; double prev_time = 0.0;
    // the GPU tracker rule's own pyramids (lk_cuda.hip: cuda::pyrDown rounds half to even): levels 1.. of the left frames / the right frame; level 0 aliases the frame
    // in left[] / right.  Built only in the modes that use that rule (naive: temporal + right image; semantic: right image)
    PyrSet leftc[2], rightc; bool leftc_valid[2] = { false, false };
    DevBuf state_block; DvTrackState tr{};
    DevBuf cand_buf; int cand_cap = 0; int* n_cand = nullptr; unsigned* max_ord = nullptr; int* err_flag = nullptr;
    DevBuf hw_buf; int hw_radius = -1;
    DevBuf mask_buf;
    // dv_track_unmask_static: the ROI masks of the static instances of the next frame, staged in pinned memory at the call (rect + offset into unmask_pinned), applied by a
    // kernel behind the mask's upload in dv_track_stereo_enqueue
    struct UnmaskJob { int x, y, w, h; size_t off; };
    std::vector<UnmaskJob> unmask; void* unmask_pinned = nullptr; size_t unmask_pinned_bytes = 0;
    DevBuf undist_buf[2]; bool undist[2] = { false, false }; int undist_w = 0, undist_h = 0;      // cfg::is_undistort_input: fixed-point maps per camera (map1 | map2)
    DevBuf out_buf; dv_feat* out_dev = nullptr; int* nout_dev = nullptr;
    dv_feat* out_pinned = nullptr; int* nout_pinned = nullptr; int* err_pinned = nullptr;
    hipEvent_t done = nullptr; bool pending = false;
    // a frame tracked as a member of a dv_batch (dv_batch_track_enqueue) runs on the batch's front-end stream and completes with the batch's event: last_done = the
    // event behind the last enqueued frame (own `done` or the batch's), last_front = the stream it ran on — a frame that runs on the OTHER stream waits for it first
    hipEvent_t last_done = nullptr; hipStream_t last_front = nullptr;
    // operator-level scratch
    PyrSet opA, opB; DevBuf s0, s1, s2, s3, s4;
    bool timing = false, kernel_timing = false, host_timing = false; std::deque<StageTimer> timers;   // deque: StageScope keeps pointers across emplace_back
    // A dynamic sequence runs the tracker API (thread T2) and the estimator API (T3) on ONE ctx (runner.hip): the timer table is shared between them — every walk /
    // growth of `timers` and every update of a timer's counters happens under timer_mu — and so is the error string (err_mu; dv_last_error hands out a
    // per-thread copy)
    std::mutex timer_mu, err_mu;
    // back end
    hipStream_t be_stream = nullptr; BeWork be;
    hipStream_t be_stream_own = nullptr;          // member of a dv_batch: be_stream IS the batch's stream (every BA launch and copy of the member is ordered on it); this is the ctx's own one, restored when it leaves
    ObjPending obj_pend, obj_op_pend;            // estimator's object solve / operator-level dv_obj_solve
    hipStream_t obj_stream = nullptr; DevBuf obj_buf;      // dynamic mode: the object solve runs beside the window solve
    DvDist dist;
    struct dv_batch* batch = nullptr;             // member of a batch of independent windows solved by shared launches (be_api.hip)
    dv_estimator* est = nullptr;
    struct dv_inst_tracker* inst = nullptr; hipEvent_t ev_pyr = nullptr, ev_bg_select = nullptr;      // dynamic mode: the per-object tracker (inst_track.hip)
};

void dv_set_error(dv_ctx* ctx, const std::string& msg);
// Stream of a dv_batch group.  DVINS_CU_PARTITIONS=P (experiment, off by default): group g's streams get the CU mask of partition g mod P — bits [256 g / P, 256 (g + 1) / P) of
// the 256-bit mask, i.e. (mask bit i = CU i / 8 of XCD i mod 8: scripts/dbg/cumask_probe.hip) the same 32 / P CUs of EVERY XCD — so that the groups' kernels do not
// compete for CUs with each other: a be_solve_batch workgroup needs a whole CU (157 KB of LDS, 16 waves of 128 VGPRs) and has to wait for one to drain completely.
hipError_t dv_group_stream_create(hipStream_t* s, int group_index);

#define DV_CHECK(expr)                                                                   \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dv_set_error(ctx, std::string(#expr) + ": " + hipGetErrorString(_e));        \
            return -1;                                                                   \
        }                                                                                \
    } while (0)
#define DV_FAIL(msg) do { dv_set_error(ctx, msg); return -1; } while (0)

StageTimer* dv_timer_for(dv_ctx* ctx, const char* name);
struct StageScope {
    dv_ctx* c; StageTimer* t = nullptr; hipStream_t s; size_t slot = 0;
    StageScope(dv_ctx* ctx, const char* name, hipStream_t st = nullptr) : c(ctx), s(st ? st : ctx->stream) {
        if (!c->timing) return;
        t = dv_timer_for(c, name);
        hipEvent_t first;
        {
            std::lock_guard<std::mutex> lk(c->timer_mu);
            t->stream = s;
            if (t->used == t->pool.size()) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); t->pool.push_back({ a, b }); }
            slot = t->used++;
            first = t->pool[slot].first; second = t->pool[slot].second;
        }
        (void)hipEventRecord(first, s);
    }
    ~StageScope() { if (t) (void)hipEventRecord(second, s); }
    hipEvent_t second = nullptr;
};
// host wall-clock of a scope (std::chrono), accumulated under `name` next to the event timers ("h_*" names)
struct HostScope {
    dv_ctx* c; StageTimer* t = nullptr; std::chrono::steady_clock::time_point t0;
    HostScope(dv_ctx* ctx, const char* name) : c(ctx) { if (!c->timing && !c->host_timing) return; t = dv_timer_for(c, name); t0 = std::chrono::steady_clock::now(); }
    ~HostScope() { if (t) { const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); std::lock_guard<std::mutex> lk(c->timer_mu); t->total_ms += ms; t->count++; } }
};
void dv_harvest_timers(dv_ctx* ctx, hipStream_t synced);      // harvests the timers recorded on `synced` (must be idle)
