// runner_tsan.cpp — ThreadSanitizer harness of the library's C++ host loop (dynamic_vins_amd/csrc/runner.hip compiled as plain C++ with -fsanitize=thread) on the
// stand-in C ABI of stub_abi.cpp: spin barriers, teams of host threads per dv_batch group, one host thread per group, the tracker thread (T2) + ring beside the
// estimator loop (T3) of a dynamic sequence, runs cut into several dv_runner_run calls, and the failure path (a member fails in the middle of a team round: every
// thread must leave, nobody may spin forever).  The stub's outputs are deterministic, so every layout must leave the same per-sequence logs as the single-thread
// loop; TSan reports what the bit-identity checks of tests/test_runner.py cannot see — a race that has not changed a result yet.
//   runner_tsan raw | dynamic | fail        exit 0 = logs identical / failure reported without a hang; TSan's own exit code (66) on a report
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "dvins.h"

extern "C" dv_ctx* dvstub_ctx(int w, int h, int dynamic);
extern "C" long long dvstub_violations();
extern "C" void dvstub_new_run();

namespace {
struct Seq {
    std::vector<const uint8_t*> left, right; std::vector<double> times, imu_t, imu_a, imu_g;
    std::vector<const uint8_t*> masks; std::vector<const dv_inst_det*> dets; std::vector<int32_t> n_dets; dv_inst_det one_det{};
    dv_seq_input in{}; dv_seq_dynamic dyn{};
};
void make_seq(Seq& q, int frames, int id) {
    static uint8_t pixel[4096];
    for (int k = 0; k < frames; ++k) { q.left.push_back(pixel + (id * 64 + k) % 4000); q.right.push_back(pixel + (id * 64 + k + 7) % 4000); q.times.push_back(1.0 + 0.05 * k); q.masks.push_back(pixel + k % 100); q.dets.push_back(&q.one_det); q.n_dets.push_back(1); }
    q.one_det.track_id = 10; q.one_det.w = q.one_det.h = 4; q.one_det.mask = pixel;
    for (int i = 0; i < frames * 10 + 20; ++i) { q.imu_t.push_back(0.9 + 0.005 * i); for (int c = 0; c < 3; ++c) { q.imu_a.push_back(0.01 * i + c + id); q.imu_g.push_back(0.02 * i - c); } }
    q.in.left = q.left.data(); q.in.right = q.right.data(); q.in.times = q.times.data(); q.in.n_frames = frames; q.in.mem = DV_MEM_DEVICE; q.in.stride = 0; q.in.ba_stride = 1;
    q.in.imu_t = q.imu_t.data(); q.in.imu_acc = q.imu_a.data(); q.in.imu_gyr = q.imu_g.data(); q.in.n_imu = (int)q.imu_t.size();
    q.dyn.inv_mask = q.masks.data(); q.dyn.mask_mem = DV_MEM_DEVICE; q.dyn.mode = DV_MODE_SEMANTIC; q.dyn.dets = q.dets.data(); q.dyn.n_dets = q.n_dets.data();
}
struct Log { std::vector<double> frames; std::vector<unsigned long long> rows; long long iterations = 0; };
// one runner over n sequences in the given layout; `cuts` = the dv_runner_run calls; -> per-sequence logs (empty on failure)
int run_layout(int n, int frames, int group, int threads, int teams, int dynamic, int tracker_thread, const std::vector<int>& cuts, std::vector<Log>& out, bool expect_fail = false, int static_bg = 0, int ba_stride = 1) {
    std::vector<Seq> seqs(n); std::vector<dv_ctx*> ctxs; std::vector<dv_seq_input> in;
    dvstub_new_run();
    for (int i = 0; i < n; ++i) { make_seq(seqs[i], frames, i); seqs[i].in.ba_stride = ba_stride; seqs[i].dyn.static_as_background = static_bg; ctxs.push_back(dvstub_ctx(64, 48, dynamic)); in.push_back(seqs[i].in); }
    dv_runner* R = dv_runner_create(ctxs.data(), in.data(), n, group, threads);
    if (!R) { std::fprintf(stderr, "dv_runner_create failed\n"); return 2; }
    dv_runner_set(R, "teams", teams);
    if (dynamic) { dv_runner_set(R, "tracker_thread", tracker_thread); for (int i = 0; i < n; ++i) if (dv_runner_set_dynamic(R, i, &seqs[i].dyn)) { std::fprintf(stderr, "set_dynamic: %s\n", dv_runner_error(R)); return 2; } }
    int rc = 0;
    for (int c : cuts) if ((rc = dv_runner_run(R, c, nullptr)) != 0) break;
    if (expect_fail) { dv_runner_destroy(R); return rc ? 0 : 3; }
    if (rc) { std::fprintf(stderr, "dv_runner_run: %s\n", dv_runner_error(R)); dv_runner_destroy(R); return 2; }
    out.assign(n, Log{});
    for (int i = 0; i < n; ++i) {
        out[i].frames.resize(9 * (size_t)frames); int nf = 0; dv_runner_get_frames(R, i, out[i].frames.data(), frames, &nf); out[i].frames.resize(9 * (size_t)nf);
        out[i].rows.resize(4 * (size_t)frames); int nr = 0; dv_runner_get_row_log(R, i, out[i].rows.data(), frames, &nr); out[i].rows.resize(4 * (size_t)nr);
        long long fr = 0; dv_runner_get(R, i, nullptr, nullptr, 0, nullptr, &out[i].iterations, &fr, nullptr);
    }
    dv_runner_destroy(R);
    return 0;
}
bool same(const std::vector<Log>& a, const std::vector<Log>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) if (a[i].frames != b[i].frames || a[i].rows != b[i].rows || a[i].iterations != b[i].iterations || a[i].frames.empty()) return false;
    return true;
}
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "raw";
    std::thread watchdog([] { std::this_thread::sleep_for(std::chrono::seconds(240)); std::fprintf(stderr, "runner_tsan: HANG (watchdog)\n"); std::_Exit(9); });
    watchdog.detach();
    int bad = 0;
    if (mode == "raw") {
        const int n = 8, frames = 36;
        std::vector<Log> ref, got;
        if (run_layout(n, frames, 4, 1, 0, 0, 0, { frames }, ref)) return 2;                                  // two dv_batch groups of four on one thread: the reference
        struct L { int group, threads, teams; std::vector<int> cuts; const char* name; };
        const L layouts[] = { { 4, 2, 0, { frames }, "one thread per group" }, { 4, 4, 1, { frames }, "teams of two" }, { 4, 8, 1, { 7, 1, 13, 15 }, "teams of four, four calls" },
                              { 0, 4, 0, { 20, 16 }, "no batching, four threads" }, { 4, 4, 1, { 1, 1, 1, 33 }, "teams of two, one-frame calls" } };
        for (const L& l : layouts) {
            if (run_layout(n, frames, l.group, l.threads, l.teams, 0, 0, l.cuts, got)) return 2;
            const bool ok = l.group == 0 ? got.size() == ref.size() : same(ref, got);      // (without dv_batch groups the stub sees other call interleavings per ctx but the same per-sequence calls)
            if (l.group == 0) { std::vector<Log> solo; if (run_layout(n, frames, 0, 1, 0, 0, 0, l.cuts, solo)) return 2; if (!same(solo, got)) bad++, std::fprintf(stderr, "MISMATCH: %s\n", l.name); }
            else if (!ok) bad++, std::fprintf(stderr, "MISMATCH: %s\n", l.name);
            std::printf("layout '%s': %s\n", l.name, ok ? "same logs" : "DIFFERENT");
        }
    } else if (mode == "dynamic") {
        const int n = 3, frames = 40;
        std::vector<Log> ref, got;
        if (run_layout(n, frames, 0, 1, 0, 1, 0, { frames }, ref)) return 2;                                  // the one-thread loop (tracker_thread 0)
        struct L { int threads, tracker; std::vector<int> cuts; const char* name; };
        const L layouts[] = { { 1, 1, { frames }, "T2 beside T3" }, { 3, 1, { frames }, "T2 beside T3, one estimator thread per sequence" }, { 1, 1, { 7, 1, 13, 19 }, "T2 beside T3, four calls" },
                              { 3, 0, { 11, 29 }, "one-thread loops on three threads" } };
        for (const L& l : layouts) {
            if (run_layout(n, frames, 0, l.threads, 0, 1, l.tracker, l.cuts, got)) return 2;
            const bool ok = same(ref, got);
            if (!ok) bad++;
            std::printf("layout '%s': %s\n", l.name, ok ? "same logs" : "DIFFERENT");
        }
        // the static-instance feedback (T3 -> T2 with a lag of two frames) and the every-2nd-frame flow: the tracker waits for the estimator's snapshot, the hand-over must not depend on the layout
        for (int stride = 1; stride <= 2; ++stride) {
            if (run_layout(n, frames, 0, 1, 0, 1, 0, { frames }, ref, false, 1, stride)) return 2;
            for (const L& l : layouts) {
                if (run_layout(n, frames, 0, l.threads, 0, 1, l.tracker, l.cuts, got, false, 1, stride)) return 2;
                const bool ok = same(ref, got);
                if (!ok) bad++;
                std::printf("static feedback, ba_stride %d, layout '%s': %s\n", stride, l.name, ok ? "same logs" : "DIFFERENT");
            }
        }
    } else if (mode == "fail") {          // DVSTUB_FAIL=<ctx>:<frame> is set by the caller: the run must return an error, not hang, in every layout
        std::vector<Log> got;
        const int frames = 30;
        if (run_layout(8, frames, 4, 4, 1, 0, 0, { frames }, got, true)) { bad++; std::fprintf(stderr, "teams: the injected failure was not reported\n"); }
        if (run_layout(8, frames, 4, 2, 0, 0, 0, { frames }, got, true)) { bad++; std::fprintf(stderr, "thread per group: the injected failure was not reported\n"); }
        if (run_layout(2, frames, 0, 1, 0, 1, 1, { frames }, got, true)) { bad++; std::fprintf(stderr, "dynamic: the injected failure was not reported\n"); }
        std::printf("failure path: %s\n", bad ? "BROKEN" : "every layout returned the error");
    } else return 2;
    if (dvstub_violations() && mode != "fail") { std::fprintf(stderr, "stub: %lld call-sequence violations\n", dvstub_violations()); bad++; }
    return bad ? 1 : 0;
}
