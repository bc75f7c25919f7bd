// stub_abi.cpp — a STAND-IN for libdvins_hip.so's C ABI for the ThreadSanitizer builds of the host-side machinery (tests/host/Makefile `tsan`): no GPU, no HIP
// runtime under TSan.  Every entry point the C++ runner (dynamic_vins_amd/csrc/runner.hip) and the header shim's concurrent surface (host/dvins_shim.hpp) call is
// here, doing three things instead of launching kernels:
//   * a state machine per context that refuses a call sequence the real library would refuse (enqueue twice, collect without enqueue, end without begin ...);
//   * plain (non-atomic) scratch words per context DOMAIN — tracker, estimator, IMU buffer — written by every call that touches that domain in the real library:
//     two host threads inside the same domain of one context without a happens-before edge are a data race TSan reports (the ABI's rule: "one ctx per thread",
//     relaxed by runner.hip to "tracker API on T2, estimator API on T3" for a dynamic sequence);
//   * deterministic outputs (functions of the frame index and of what the call was handed) after a short sleep, so that every threading layout of the runner must
//     produce the same logs as its single-thread loop.
// DVSTUB_FAIL="<ctx index>:<frame>" makes dv_est_process_begin of that context fail at that frame: the failure path through the team barriers.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "dv_ctx.h"

namespace {
struct Stub {
    int index = 0;
    // tracker domain
    int trk_scratch = 0; bool trk_pending = false; int trk_frame = 0; double trk_t = 0; unsigned long long trk_in = 0, unmask_hash = 0;
    bool inst_pending = false; int inst_frame = 0;
    // estimator domain
    int est_scratch = 0; bool begun = false, ego_begun = false; int est_frame = 0; unsigned long long est_hash = 0; double est_t = 0;
    // IMU-buffer domain
    int imu_scratch = 0; long long imu_n = 0; double imu_sum = 0, imu_last_t = -1;
};
std::mutex g_mu; std::unordered_map<dv_ctx*, Stub*> g_map; std::vector<dv_ctx*> g_order;
std::string g_err; std::atomic<long long> g_violations{0};
int g_fail_ctx = -1, g_fail_frame = -1, g_next_index = 0;

Stub& S(dv_ctx* c) { return *g_map.at(c); }      // (no lock: the map is complete before any thread starts, and a lock here would order the very accesses TSan is to judge)
void work(int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }
int violation(const char* what) { g_violations.fetch_add(1); std::lock_guard<std::mutex> lk(g_mu); g_err = what; return -1; }
unsigned long long mix(unsigned long long h, unsigned long long v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); return h; }
unsigned long long bits(double d) { unsigned long long u; std::memcpy(&u, &d, 8); return u; }
}

// GCC 11's libtsan has no interceptor for pthread_cond_clockwait (what std::condition_variable::wait_for / wait_until(steady_clock) call on glibc >= 2.30): TSan then
// never sees the mutex released inside the wait and reports "double lock" and phantom races on everything the mutex guards.  The harness binaries route the call
// through pthread_cond_timedwait, which IS intercepted (test infrastructure only; the product is not built with this file).
#if defined(__SANITIZE_THREAD__)
#include <pthread.h>
#include <time.h>
extern "C" int pthread_cond_clockwait(pthread_cond_t* cond, pthread_mutex_t* mutex, clockid_t clock, const struct timespec* abstime) {
    struct timespec now_c, now_r, abs_r;
    clock_gettime(clock, &now_c); clock_gettime(CLOCK_REALTIME, &now_r);
    long long rel = (abstime->tv_sec - now_c.tv_sec) * 1000000000ll + (abstime->tv_nsec - now_c.tv_nsec);
    if (rel < 0) rel = 0;
    const long long t = now_r.tv_sec * 1000000000ll + now_r.tv_nsec + rel;
    abs_r.tv_sec = t / 1000000000ll; abs_r.tv_nsec = t % 1000000000ll;
    return pthread_cond_timedwait(cond, mutex, &abs_r);
}
#endif

void dv_set_error(dv_ctx*, const std::string& m) { std::lock_guard<std::mutex> lk(g_mu); g_err = m; }

extern "C" {
long long dvstub_violations() { return g_violations.load(); }
void dvstub_new_run() { g_next_index = 0; }          // the contexts created from here on are numbered from 0 again (DVSTUB_FAIL addresses them per run)
dv_ctx* dvstub_ctx(int w, int h, int dynamic) {          // what dv_create + dv_est_create (+ dv_inst_config) leave as far as runner.hip looks
    static int dummy_est, dummy_inst;
    if (const char* e = std::getenv("DVSTUB_FAIL")) std::sscanf(e, "%d:%d", &g_fail_ctx, &g_fail_frame);
    dv_ctx* c = new dv_ctx();
    c->cfg.width = w; c->cfg.height = h;
    c->est = reinterpret_cast<dv_estimator*>(&dummy_est);
    if (dynamic) c->inst = reinterpret_cast<dv_inst_tracker*>(&dummy_inst);
    std::lock_guard<std::mutex> lk(g_mu);
    Stub* s = new Stub(); s->index = g_next_index++; g_map[c] = s; g_order.push_back(c);
    return c;
}
const char* dv_last_error(dv_ctx*) { static thread_local std::string mine; std::lock_guard<std::mutex> lk(g_mu); mine = g_err; return mine.c_str(); }
int dv_sync(dv_ctx*) { return 0; }

// ---------------- tracker domain ----------------
int dv_track_stereo_enqueue(dv_ctx* c, const uint8_t* g0, const uint8_t* g1, int, int, int, double t, const uint8_t* mask, int mode, int) {
    Stub& s = S(c); s.trk_scratch++;
    if (s.trk_pending) return violation("dv_track_stereo_enqueue: previous frame not collected");
    s.trk_pending = true; s.trk_t = t; s.trk_in = mix(mix(mix(mix((unsigned long long)(uintptr_t)g0, (unsigned long long)(uintptr_t)g1), (unsigned long long)(uintptr_t)mask), (unsigned long long)mode), s.unmask_hash);
    s.unmask_hash = 0;
    work(30);
    return 0;
}
int dv_track_stereo_collect(dv_ctx* c, dv_feat* out, int* n_out) {
    Stub& s = S(c); s.trk_scratch++;
    if (!s.trk_pending) return violation("dv_track_stereo_collect: nothing enqueued");
    work(60);
    const int n = 5 + s.trk_frame % 7;
    for (int i = 0; i < n; ++i) { dv_feat f{}; f.id = (uint32_t)(s.trk_frame * 100 + i); f.track_cnt = s.trk_frame + 1; f.left[0] = s.trk_t; f.left[1] = (double)(s.trk_in & 0xffff); f.left[3] = i; out[i] = f; }
    if (n_out) *n_out = n;
    s.trk_pending = false; s.trk_frame++;
    return 0;
}
int dv_inst_set_disparity(dv_ctx* c, const float*, int, int, double) { Stub& s = S(c); s.trk_scratch++; return 0; }
int dv_inst_set_right_keys(dv_ctx* c, const uint32_t*, int, int) { Stub& s = S(c); s.trk_scratch++; return 0; }
int dv_track_unmask_static(dv_ctx* c, const dv_inst_det*, int, const uint32_t* ids, int n) {          // the ids travel into the next frame's input hash: every layout must hand over the same list
    Stub& s = S(c); s.trk_scratch++;
    if (s.trk_pending) return violation("dv_track_unmask_static: behind the frame's enqueue");
    s.unmask_hash = 0x51ed; for (int i = 0; i < n; ++i) s.unmask_hash = mix(s.unmask_hash, ids[i]);
    return 0;
}
int dv_inst_track_enqueue(dv_ctx* c, double, const dv_inst_det*, int, const dv_box3d*, int) {
    Stub& s = S(c); s.trk_scratch++;
    if (s.inst_pending) return violation("dv_inst_track_enqueue: previous frame not collected");
    s.inst_pending = true; work(20);
    return 0;
}
int dv_inst_track_collect(dv_ctx* c, dv_inst_obs* insts, int cap_insts, int* n_insts, dv_feat* feats, int cap_feats, int* n_feats, double* points, int cap_points, int* n_points) {
    Stub& s = S(c); s.trk_scratch++;
    if (!s.inst_pending) return violation("dv_inst_track_collect: nothing enqueued");
    work(40);
    const int ni = 1 + s.inst_frame % 3, per = 4;
    if (ni > cap_insts || ni * per > cap_feats || ni * 2 > cap_points) return violation("dv_inst_track_collect: buffers too small");
    for (int i = 0; i < ni; ++i) { dv_inst_obs o{}; o.id = (uint32_t)(10 + i); o.first_feat = i * per; o.n_feats = per; o.first_point = 2 * i; o.n_points = 2; insts[i] = o; }
    for (int i = 0; i < ni * per; ++i) { dv_feat f{}; f.id = (uint32_t)(s.inst_frame * 1000 + i); f.left[0] = s.inst_frame; feats[i] = f; }
    for (int i = 0; i < ni * 2 * 3; ++i) points[i] = s.inst_frame + 0.25 * i;
    *n_insts = ni; *n_feats = ni * per; *n_points = ni * 2;
    s.inst_pending = false; s.inst_frame++;
    return 0;
}

// ---------------- IMU buffer + estimator domains ----------------
int dv_est_input_imu(dv_ctx* c, double t, const double* acc, const double* gyr) {
    Stub& s = S(c); s.imu_scratch++;
    if (t <= s.imu_last_t) return violation("dv_est_input_imu: samples out of order");
    s.imu_last_t = t; s.imu_n++; s.imu_sum += acc[0] + 2 * acc[1] + 3 * acc[2] + 5 * gyr[0] + 7 * gyr[1] + 11 * gyr[2];
    return 0;
}
int dv_est_imu_available(dv_ctx* c, double t) { Stub& s = S(c); s.imu_scratch++; return s.imu_last_t >= t ? 1 : 0; }
int dv_est_get_latest(dv_ctx* c, double* t, double* P, double* Q, double* V) { Stub& s = S(c); s.imu_scratch++; if (t) *t = s.imu_last_t; for (int k = 0; k < 3; ++k) { P[k] = s.imu_sum; V[k] = 0; } Q[0] = Q[1] = Q[2] = 0; Q[3] = 1; return s.imu_n ? 0 : 1; }
static int begin_common(Stub& s, const dv_feat* rows, int n, double t) {
    s.est_scratch++; s.imu_scratch++;
    if (s.begun) return violation("dv_est_process_begin: previous frame not ended");
    if (s.index == g_fail_ctx && s.est_frame == g_fail_frame) { std::lock_guard<std::mutex> lk(g_mu); g_err = "stub: injected failure"; return -1; }
    unsigned long long h = mix(bits(t), (unsigned long long)n);
    for (int i = 0; i < n; ++i) h = mix(mix(h, rows[i].id), bits(rows[i].left[0]));
    h = mix(mix(h, (unsigned long long)s.imu_n), bits(s.imu_sum));
    s.est_hash = h; s.est_t = t; s.begun = true;
    work(50);
    return 0;
}
int dv_est_process_begin(dv_ctx* c, const dv_feat* rows, int n, double t) { return begin_common(S(c), rows, n, t); }
int dv_est_process_dynamic_begin_ego(dv_ctx* c, const dv_feat* rows, int n, double t) { Stub& s = S(c); const int rc = begin_common(s, rows, n, t); if (!rc) s.ego_begun = true; return rc; }
int dv_est_process_dynamic_attach(dv_ctx* c, const dv_inst_obs* insts, int n_insts, const dv_feat* feats, const double* points) {
    Stub& s = S(c); s.est_scratch++;
    if (!s.ego_begun) return violation("dv_est_process_dynamic_attach: no ego solve in flight");
    s.ego_begun = false;
    for (int i = 0; i < n_insts; ++i) { s.est_hash = mix(s.est_hash, insts[i].id); for (int k = 0; k < insts[i].n_feats; ++k) s.est_hash = mix(s.est_hash, feats[insts[i].first_feat + k].id); for (int k = 0; k < 3 * insts[i].n_points; ++k) s.est_hash = mix(s.est_hash, bits(points[3 * insts[i].first_point + k])); }
    work(40);
    return 0;
}
int dv_est_process_end(dv_ctx* c, dv_est_state* out) {
    Stub& s = S(c); s.est_scratch++;
    if (!s.begun) return violation("dv_est_process_end: nothing begun");
    if (s.ego_begun) return violation("dv_est_process_end: the object branch was not attached");
    work(80);
    dv_est_state st{};
    st.frame = s.est_frame; st.nonlinear = s.est_frame >= 3; st.iterations = 1 + (int)(s.est_hash % 5);
    for (int k = 0; k < 7; ++k) st.window[10][k] = (double)((s.est_hash >> (8 * k)) & 0xff) + s.est_t;
    if (out) *out = st;
    s.begun = false; s.est_frame++;
    return 0;
}
int dv_est_process(dv_ctx* c, const dv_feat* rows, int n, double t, dv_est_state* out) {          // the shim's one-call form
    Stub& s = S(c);
    if (s.imu_last_t < t) { s.imu_scratch++; return 1; }          // the IMU stream does not cover the frame yet
    if (begin_common(s, rows, n, t)) return -1;
    return dv_est_process_end(c, out);
}
int dv_est_set_lines(dv_ctx* c, const dv_line_row*, int) { S(c).est_scratch++; return 0; }
int dv_est_get_static_instances(dv_ctx* c, uint32_t* ids, int cap, int* n) {          // "static" instances as a function of the frames processed so far
    Stub& s = S(c); s.est_scratch++;
    const int k = std::min(cap, s.est_frame % 3);
    for (int i = 0; i < k; ++i) ids[i] = (uint32_t)(10 + (s.est_frame + i) % 3);
    *n = k;
    return 0;
}

// ---------------- dv_batch: the shared launches touch every member ----------------
struct dv_batch { std::vector<dv_ctx*> m; long long rounds = 0, track_rounds = 0, members = 0; };
dv_batch* dv_batch_create(dv_ctx* const* ctxs, int n) { dv_batch* b = new dv_batch(); b->m.assign(ctxs, ctxs + n); return b; }
void dv_batch_destroy(dv_batch* b) { delete b; }
int dv_batch_enqueue(dv_batch* b) {
    for (dv_ctx* c : b->m) { Stub& s = S(c); s.est_scratch++; if (s.begun == false) { /* a member without a frame this round: allowed */ } }
    b->rounds++; work(40);
    return 0;
}
int dv_batch_track_enqueue(dv_batch* b, const dv_track_job* jobs, int n) {
    for (int i = 0; i < n; ++i) {
        if (jobs[i].member < 0 || jobs[i].member >= (int)b->m.size()) return violation("dv_batch_track_enqueue: member index out of range");
        if (dv_track_stereo_enqueue(b->m[jobs[i].member], jobs[i].gray0, jobs[i].gray1, 0, 0, 0, jobs[i].t, jobs[i].mask, jobs[i].mode, jobs[i].mem)) return -1;
    }
    b->track_rounds++; b->members += n;
    return 0;
}
int dv_batch_info(dv_batch* b, long long* a, long long* s) { if (a) *a = b->rounds; if (s) *s = 0; return 0; }
int dv_batch_track_info(dv_batch* b, long long* r, long long* mb, long long* ms) { if (r) *r = b->track_rounds; if (mb) *mb = b->members; if (ms) *ms = 0; return 0; }
int dv_batch_timing(dv_batch*, int, double* o, long long* r, int* w) { if (o) o[0] = o[1] = o[2] = 0; if (r) *r = 0; if (w) *w = 0; return 0; }

// ---------------- what the header shim's classes call besides the above (host/dvins_shim.hpp): construction, the one-call forms, the getters ----------------
dv_ctx* dv_create(const dv_config* cfg) { dv_ctx* c = dvstub_ctx(cfg ? cfg->width : 64, cfg ? cfg->height : 48, 0); if (cfg) c->cfg = *cfg; return c; }
void dv_destroy(dv_ctx*) {}
int dv_est_create(dv_ctx* c, const dv_est_config*) { S(c).est_scratch++; return 0; }
int dv_est_reset(dv_ctx* c) { Stub& s = S(c); s.est_scratch++; s.imu_scratch++; s.begun = s.ego_begun = false; s.est_frame = 0; s.imu_n = 0; s.imu_sum = 0; s.imu_last_t = -1; return 0; }
int dv_est_change_sensor_type(dv_ctx* c, int, int) { S(c).est_scratch++; return 0; }
int dv_est_get_lines(dv_ctx* c, dv_line_landmark*, int, int* n) { S(c).est_scratch++; if (n) *n = 0; return 0; }
int dv_est_get_landmarks(dv_ctx* c, dv_landmark*, int, int* n) { S(c).est_scratch++; if (n) *n = 0; return 0; }
int dv_est_get_instances(dv_ctx* c, dv_inst_state*, int, int* n, double*) { S(c).est_scratch++; if (n) *n = 0; return 0; }
int dv_est_process_dynamic(dv_ctx* c, const dv_feat* rows, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* feats, const double* points, dv_est_state* out) {
    Stub& s = S(c);
    if (s.imu_last_t < t) { s.imu_scratch++; return 1; }
    if (dv_est_process_dynamic_begin_ego(c, rows, n, t) || dv_est_process_dynamic_attach(c, insts, n_insts, feats, points)) return -1;
    return dv_est_process_end(c, out);
}
int dv_track_stereo(dv_ctx* c, const uint8_t* g0, const uint8_t* g1, int w, int h, int stride, double t, const uint8_t* mask, int mode, int mem, dv_feat* out, int* n_out) {
    if (dv_track_stereo_enqueue(c, g0, g1, w, h, stride, t, mask, mode, mem)) return -1;
    return dv_track_stereo_collect(c, out, n_out);
}
void* dv_pinned_alloc(size_t n) { return std::malloc(n); }
void dv_pinned_free(void* p) { std::free(p); }
int dv_inst_config(dv_ctx* c, int, int, int) { S(c).trk_scratch++; return 0; }
int dv_undistort_lines(dv_ctx*, const dv_cam*, const float*, int, double*) { return 0; }
int dv_set_undistort_maps(dv_ctx* c, int, const int16_t*, const uint16_t*, int, int) { S(c).trk_scratch++; return 0; }
int dv_remap(dv_ctx* c, const uint8_t*, int, int, int, int, const int16_t*, const uint16_t*, uint8_t*, int) { S(c).trk_scratch++; return 0; }
int dv_obj_solve(dv_ctx* c, dv_obj_problem*, dv_ba_summary*) { S(c).est_scratch++; return 0; }
}
