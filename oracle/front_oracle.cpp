// front_oracle.cpp — CPU ORACLE (test infrastructure, not the product) for the front end.
//
// Restates, dependency-free:
//   * OpenCV 3.4.16 calcOpticalFlowPyrLK / buildOpticalFlowPyramid / calcSharrDeriv / pyrDown
//     (modules/video/src/lkpyramid.cpp, modules/imgproc/src/pyramids.cpp) — un-vendored
//     third-party code, restated from the published algorithm;
//   * OpenCV 3.4.16 goodFeaturesToTrack / cornerMinEigenVal (featureselect.cpp, corner.cpp),
//     circle() (drawing.cpp, Circle()), erode (App. A.2 / A.5);
//   * camodocal PinholeCamera::liftProjective (camera_models/src/camera_models/PinholeCamera.cc:450-508,646-662);
//   * dynamic_vins FeatureTrackByLK (front_end/feature_utils.cpp:35-69), InBorder / ReduceVector
//     (feature_utils.h:68-85), SortPoints (feature_utils.cpp:307-328), InstFeat::PtsVelocity
//     (front_end/instance_feature.cpp:26-85), FeatureTracker::TrackImage / TrackImageNaive /
//     SetOutputFeats (front_end/background_tracker.cpp:52-158, 340-392, 400-516).
//
// PARITY UNPINNED (see dvo.h).  Documented canonical choices where OpenCV itself is
// platform dependent:
//   D1  LK window sums (A11,A12,A22,b1,b2) are sums of integers that OpenCV accumulates in
//       float in a platform-dependent order (scalar / SSE2 4-lane / NEON).  The oracle takes
//       the exact integer sum (int64) and converts to float once.
//   D2  boxFilter on the CV_32FC3 covariance image uses double sliding sums in OpenCV; the
//       oracle sums the 3x3 window in double in a fixed order ((l+c)+r per row, (t+m)+b).
//   D3  SortPoints uses unstable std::sort on track_cnt (Q11); the oracle uses a stable sort.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off (no FMA contraction: results must not depend
// on the host ISA).
#include "dvo.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include <string>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
static int g_threads = 1;      // dvo_set_threads: CPU-baseline timing only; results do not depend on it (rows / points are independent, lists are concatenated in order)
extern "C" void dvo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
extern "C" int dvo_get_threads() { return g_threads; }
// sensitivity variants (dvo.h): process-wide switches, read by the stages they belong to
int g_var_lk_sums = 0, g_var_box_sums = 0, g_var_radius = 0, g_var_f4_cpu = 0, g_var_f5_cpu = 0, g_var_gc_fma = 0, g_var_gc_tie = 0, g_var_obj_order = 0, g_var_obj_perturb = 0, g_var_obj_dump = 0;      // "f4_cpu_rule" 1: naive / semantic modes track with the CPU arithmetic + the 1.0 px threshold (the substitution of rounds 1-3) instead of the GPU tracker (lk_cuda.cpp)
extern "C" void dvo_set_variant(const char* key, int value) {
    const std::string k = key ? key : "";
    if (k == "lk_sums") g_var_lk_sums = value; else if (k == "box_sums") g_var_box_sums = value; else if (k == "radius") g_var_radius = value; else if (k == "f4_cpu_rule") g_var_f4_cpu = value;
    else if (k == "f5_cpu_rule") g_var_f5_cpu = value; else if (k == "gftt_cuda_fma") g_var_gc_fma = value; else if (k == "gftt_cuda_tie") g_var_gc_tie = value;      // gftt_cuda.cpp
    else if (k == "obj_point_order") g_var_obj_order = value;      // obj_solve.cpp
    else if (k == "obj_perturb") g_var_obj_perturb = value;        // inst_manager.h
    else if (k == "obj_dump") g_var_obj_dump = value;              // inst_manager.h (a test hook, not a variant of the arithmetic)
}
extern "C" int dvo_get_variant(const char* key) {
    const std::string k = key ? key : "";
    return k == "lk_sums" ? g_var_lk_sums : k == "box_sums" ? g_var_box_sums : k == "radius" ? g_var_radius : k == "f4_cpu_rule" ? g_var_f4_cpu :
           k == "f5_cpu_rule" ? g_var_f5_cpu : k == "gftt_cuda_fma" ? g_var_gc_fma : k == "gftt_cuda_tie" ? g_var_gc_tie : k == "obj_point_order" ? g_var_obj_order : k == "obj_perturb" ? g_var_obj_perturb : k == "obj_dump" ? g_var_obj_dump : -1;
}
// A persistent worker pool standing in for OpenCV's parallel_for_ back end (the reference links OpenCV 3.4 built with a thread pool: calcOpticalFlowPyrLK runs
// its LKTrackerInvoker over ranges of points, pyrDown / the corner response over ranges of rows).  Spawning std::threads per level — the round-2 form — cost
// more than the work it split (16 threads = 0 % gain).  parallel_rows(n, grain, f) calls f(a, b) on disjoint ranges covering [0, n).
namespace {
class Pool {
public:
    static Pool& get() { static Pool* p = new Pool; return *p; }      // never destroyed: its workers sleep on the condition variable until the process ends
    void run(int n, int grain, const std::function<void(int, int)>& f) {
        const int want = std::min(g_threads, std::max(1, n / std::max(grain, 1)));
        if (want <= 1 || busy_) { f(0, n); return; }                 // (nested calls run inline)
        ensure(want - 1);
        std::unique_lock<std::mutex> lk(mu_);
        busy_ = true; fn_ = &f; n_ = n; parts_ = want; next_ = 0; pending_ = want; ++gen_;
        lk.unlock(); cv_.notify_all();
        work();                                                      // the caller takes parts too
        lk.lock();
        done_.wait(lk, [&] { return pending_ == 0; });
        busy_ = false; fn_ = nullptr;
    }
private:
    void ensure(int workers) {
        while ((int)th_.size() < workers) { th_.emplace_back([this] { loop(); }); th_.back().detach(); }
    }
    void work() {
        for (;;) {
            int part;
            { std::lock_guard<std::mutex> lk(mu_); if (next_ >= parts_) return; part = next_++; }
            const int a = (int)((long long)n_ * part / parts_), b = (int)((long long)n_ * (part + 1) / parts_);
            (*fn_)(a, b);
            { std::lock_guard<std::mutex> lk(mu_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    void loop() {
        long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; }
            work();
        }
    }
    std::mutex mu_; std::condition_variable cv_, done_;
    std::vector<std::thread> th_;              // detached at exit by leaking the singleton's threads (process lifetime)
    const std::function<void(int, int)>* fn_ = nullptr;
    int n_ = 0, parts_ = 0, next_ = 0, pending_ = 0; long long gen_ = 0; bool busy_ = false;
};
inline void parallel_rows(int n, int grain, const std::function<void(int, int)>& f) { Pool::get().run(n, grain, f); }
}

namespace {

inline int reflect101(int p, int len) {          // cv::borderInterpolate(BORDER_REFLECT_101)
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}
inline int cv_round(double v) { return (int)std::nearbyint(v); }   // round-half-even (default FE mode)
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }   // CV_DESCALE

struct Img { int w = 0, h = 0; std::vector<uint8_t> d; };

void pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
    const int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    parallel_rows(h2, 32, [&](int ya, int yb) {
        // horizontal [1 4 6 4 1] pass of a source row, cached in a ring of 5 (consecutive output rows share three of their five source rows)
        std::vector<int> ring(5 * (size_t)w2);
        int have[5] = { -1, -1, -1, -1, -1 };
        auto hrow = [&](int sy) -> const int* {
            const int slot = sy % 5;
            int* r = ring.data() + (size_t)slot * w2;
            if (have[slot] == sy) return r;
            have[slot] = sy;
            const uint8_t* s = src + (size_t)sy * w;
            for (int x = 0; x < w2; ++x) {
                if (x >= 1 && 2 * x + 2 < w) { const uint8_t* q = s + 2 * x; r[x] = q[-2] + q[2] + 4 * (q[-1] + q[1]) + 6 * q[0]; continue; }      // interior: no border arithmetic
                int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w), x2 = 2 * x < w ? 2 * x : reflect101(2 * x, w);
                int x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                r[x] = s[x0] + s[x4] + 4 * (s[x1] + s[x3]) + 6 * s[x2];
            }
            return r;
        };
        for (int y = ya; y < yb; ++y) {
            // the five (reflected) source rows of an output row lie within five consecutive row indices, so they never share a ring slot
            const int* r[5];
            for (int k = 0; k < 5; ++k) r[k] = hrow(reflect101(2 * y + k - 2, h));
            for (int x = 0; x < w2; ++x) {
                int v = r[0][x] + r[4][x] + 4 * (r[1][x] + r[3][x]) + 6 * r[2][x];
                dst[(size_t)y * w2 + x] = (uint8_t)((v + 128) >> 8);
            }
        }
    });
}

// calcSharrDeriv: Ix = [3 10 3]^T (vertical smooth) x [-1 0 1]; Iy = [-1 0 1]^T x [3 10 3]; REFLECT_101
void scharr(const uint8_t* src, int w, int h, int16_t* out) {
  parallel_rows(h, 64, [&](int ya, int yb) {
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = ya; y < yb; ++y) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
        const uint8_t* r1 = src + (size_t)y * w;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
        for (int x = 0; x < w; ++x) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int xl = w > 1 ? 1 : 0, xr = w > 1 ? w - 2 : 0;
        t0[0] = t0[xl + 1]; t0[w + 1] = t0[xr + 1];
        t1[0] = t1[xl + 1]; t1[w + 1] = t1[xr + 1];
        for (int x = 0; x < w; ++x) {
            out[((size_t)y * w + x) * 2 + 0] = (int16_t)(t0[x + 2] - t0[x]);
            out[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
  });
}

constexpr int WIN = 21;     // cv::Size(21,21) at every call site (feature_utils.cpp:44,51)
constexpr int PAD = WIN;    // buildOpticalFlowPyramid pads each level by winSize

struct Level {
    int w, h;
    std::vector<uint8_t> img;     // (w+2*PAD) x (h+2*PAD), REFLECT_101 border
    std::vector<int16_t> der;     // same size x2, CONSTANT(0) border
    int stride() const { return w + 2 * PAD; }
    const uint8_t* I(int x, int y) const { return img.data() + (size_t)(y + PAD) * stride() + (x + PAD); }
    const int16_t* D(int x, int y) const { return der.data() + ((size_t)(y + PAD) * stride() + (x + PAD)) * 2; }
};

// buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, false, REFLECT_101, CONSTANT)
int build_pyramid(const uint8_t* img, int w, int h, int max_level, std::vector<Level>& pyr, bool with_deriv) {
    pyr.clear();
    std::vector<uint8_t> cur(img, img + (size_t)w * h);
    int cw = w, ch = h, level = 0;
    for (;; ++level) {
        Level L; L.w = cw; L.h = ch;
        const int st = L.stride();
        L.img.resize((size_t)st * (ch + 2 * PAD));
        for (int y = -PAD; y < ch + PAD; ++y) {
            const uint8_t* s = cur.data() + (size_t)reflect101(y, ch) * cw;
            uint8_t* d = L.img.data() + (size_t)(y + PAD) * st;
            for (int x = -PAD; x < cw + PAD; ++x) d[x + PAD] = s[reflect101(x, cw)];
        }
        if (with_deriv) {
            L.der.assign((size_t)st * (ch + 2 * PAD) * 2, 0);
            std::vector<int16_t> d((size_t)cw * ch * 2);
            scharr(cur.data(), cw, ch, d.data());
            for (int y = 0; y < ch; ++y)
                std::memcpy(L.der.data() + ((size_t)(y + PAD) * st + PAD) * 2, d.data() + (size_t)y * cw * 2, (size_t)cw * 2 * sizeof(int16_t));
        }
        pyr.push_back(std::move(L));
        if (level == max_level) break;
        int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
        if (nw <= WIN || nh <= WIN) break;      // lkpyramid.cpp: level too small -> stop, return level
        std::vector<uint8_t> nxt((size_t)nw * nh);
        pyr_down(cur.data(), cw, ch, nxt.data());
        cur.swap(nxt); cw = nw; ch = nh;
    }
    return (int)pyr.size() - 1;
}

struct P2f { float x, y; };

// LKTrackerInvoker::operator() for one level (lkpyramid.cpp)
void lk_level(const Level& I, const Level& J, int level, int max_level, const P2f* prev_pts, P2f* next_pts,
              uint8_t* status, int n, int max_count, double eps_sq, bool use_initial) {
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int W_BITS = 14;
    int16_t Iw[WIN * WIN], dIw[WIN * WIN * 2];
    for (int p = 0; p < n; ++p) {
        P2f prevPt = { prev_pts[p].x * (float)(1. / (1 << level)), prev_pts[p].y * (float)(1. / (1 << level)) };
        P2f nextPt;
        if (level == max_level) {
            if (use_initial) nextPt = { next_pts[p].x * (float)(1. / (1 << level)), next_pts[p].y * (float)(1. / (1 << level)) };
            else nextPt = prevPt;
        } else nextPt = { next_pts[p].x * 2.f, next_pts[p].y * 2.f };
        next_pts[p] = nextPt;

        prevPt.x -= half; prevPt.y -= half;
        int ipx = cv_floor(prevPt.x), ipy = cv_floor(prevPt.y);
        if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
            if (level == 0) status[p] = 0;
            continue;
        }
        float a = prevPt.x - ipx, b = prevPt.y - ipy;
        int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        const int st = I.stride();
        int64_t sA11 = 0, sA12 = 0, sA22 = 0;      // D1: exact integer sums
        const int sum_mode = g_var_lk_sums;
        float fA11[4] = { 0, 0, 0, 0 }, fA12[4] = { 0, 0, 0, 0 }, fA22[4] = { 0, 0, 0, 0 };      // variants 1 / 2: float accumulation (lane 0 only, or x mod 4)
        for (int y = 0; y < WIN; ++y) {
            const uint8_t* src = I.I(ipx, ipy + y);
            const int16_t* ds = I.D(ipx, ipy + y);
            for (int x = 0; x < WIN; ++x) {
                int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + st] * iw10 + src[x + st + 1] * iw11, W_BITS - 5);
                int ixval = descale(ds[2 * x] * iw00 + ds[2 * x + 2] * iw01 + ds[2 * (x + st)] * iw10 + ds[2 * (x + st) + 2] * iw11, W_BITS);
                int iyval = descale(ds[2 * x + 1] * iw00 + ds[2 * x + 3] * iw01 + ds[2 * (x + st) + 1] * iw10 + ds[2 * (x + st) + 3] * iw11, W_BITS);
                Iw[y * WIN + x] = (int16_t)ival;
                dIw[(y * WIN + x) * 2] = (int16_t)ixval;
                dIw[(y * WIN + x) * 2 + 1] = (int16_t)iyval;
                sA11 += (int64_t)ixval * ixval; sA12 += (int64_t)ixval * iyval; sA22 += (int64_t)iyval * iyval;
                if (sum_mode) { const int l = sum_mode == 2 ? (x & 3) : 0; fA11[l] += (float)(ixval * ixval); fA12[l] += (float)(ixval * iyval); fA22[l] += (float)(iyval * iyval); }
            }
        }
        float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        if (sum_mode) {
            A11 = ((fA11[0] + fA11[1]) + (fA11[2] + fA11[3])) * FLT_SCALE; A12 = ((fA12[0] + fA12[1]) + (fA12[2] + fA12[3])) * FLT_SCALE; A22 = ((fA22[0] + fA22[1]) + (fA22[2] + fA22[3])) * FLT_SCALE;
        }
        float D = A11 * A22 - A12 * A12;
        float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
        if (minEig < 1e-4f || D < FLT_EPSILON) {          // minEigThreshold = 1e-4 (default)
            if (level == 0) status[p] = 0;
            continue;
        }
        D = 1.f / D;
        nextPt.x -= half; nextPt.y -= half;
        P2f prevDelta = { 0.f, 0.f };
        const int stJ = J.stride();
        for (int j = 0; j < max_count; ++j) {
            int inx = cv_floor(nextPt.x), iny = cv_floor(nextPt.y);
            if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
                if (level == 0) status[p] = 0;
                break;
            }
            a = nextPt.x - inx; b = nextPt.y - iny;
            iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t sb1 = 0, sb2 = 0;
            float fb1[4] = { 0, 0, 0, 0 }, fb2[4] = { 0, 0, 0, 0 };
            for (int y = 0; y < WIN; ++y) {
                const uint8_t* Jp = J.I(inx, iny + y);
                for (int x = 0; x < WIN; ++x) {
                    int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stJ] * iw10 + Jp[x + stJ + 1] * iw11, W_BITS - 5) - Iw[y * WIN + x];
                    sb1 += (int64_t)diff * dIw[(y * WIN + x) * 2];
                    sb2 += (int64_t)diff * dIw[(y * WIN + x) * 2 + 1];
                    if (sum_mode) { const int l = sum_mode == 2 ? (x & 3) : 0; fb1[l] += (float)(diff * dIw[(y * WIN + x) * 2]); fb2[l] += (float)(diff * dIw[(y * WIN + x) * 2 + 1]); }
                }
            }
            float b1 = (float)sb1 * FLT_SCALE, b2 = (float)sb2 * FLT_SCALE;
            if (sum_mode) { b1 = ((fb1[0] + fb1[1]) + (fb1[2] + fb1[3])) * FLT_SCALE; b2 = ((fb2[0] + fb2[1]) + (fb2[2] + fb2[3])) * FLT_SCALE; }
            P2f delta = { (float)((A12 * b2 - A22 * b1) * D), (float)((A12 * b1 - A11 * b2) * D) };
            nextPt.x += delta.x; nextPt.y += delta.y;
            next_pts[p] = { nextPt.x + half, nextPt.y + half };
            if ((double)delta.x * delta.x + (double)delta.y * delta.y <= eps_sq) break;
            if (j > 0 && std::abs(delta.x + prevDelta.x) < 0.01 && std::abs(delta.y + prevDelta.y) < 0.01) {
                next_pts[p].x -= delta.x * 0.5f; next_pts[p].y -= delta.y * 0.5f;
                break;
            }
            prevDelta = delta;
        }
        // err != NULL at both call sites -> the level-0 re-check of the final point runs
        if (status[p] && level == 0) {
            P2f np = { next_pts[p].x - half, next_pts[p].y - half };
            int ix = cv_floor(np.x), iy = cv_floor(np.y);
            if (ix < -WIN || ix >= J.w || iy < -WIN || iy >= J.h) status[p] = 0;
        }
    }
}

void lk(const uint8_t* img_a, const uint8_t* img_b, int w, int h, const P2f* pts_a, int n, int max_level,
        int iters, double eps, bool use_initial, P2f* pts_b, uint8_t* status) {
    if (n <= 0) return;
    std::vector<Level> pa, pb;
    int la = build_pyramid(img_a, w, h, max_level, pa, true);
    int lb = build_pyramid(img_b, w, h, max_level, pb, false);
    int ml = std::min(la, lb);
    iters = std::min(std::max(iters, 0), 100);
    eps = std::min(std::max(eps, 0.), 10.);
    double eps_sq = eps * eps;
    for (int i = 0; i < n; ++i) status[i] = 1;
    for (int level = ml; level >= 0; --level)          // cv::calcOpticalFlowPyrLK: parallel_for_ over the points of a level (LKTrackerInvoker)
        parallel_rows(n, 8, [&](int a, int b) { lk_level(pa[level], pb[level], level, ml, pts_a + a, pts_b + a, status + a, b - a, iters, eps_sq, use_initial); });
}

inline bool in_border(P2f pt, int rows, int cols) {      // feature_utils.h:68-74
    int x = cv_round(pt.x), y = cv_round(pt.y);
    return 1 <= x && x < cols - 1 && 1 <= y && y < rows - 1;
}

void track_by_lk(const uint8_t* img1, const uint8_t* img2, int w, int h, const P2f* pts1, int n, bool flow_back,
                 float dist_thresh, P2f* pts2, uint8_t* status) {   // feature_utils.cpp:35-69
    lk(img1, img2, w, h, pts1, n, 3, 30, 0.01, false, pts2, status);
    if (flow_back) {
        std::vector<uint8_t> rs(n);
        std::vector<P2f> rp(pts1, pts1 + n);
        lk(img2, img1, w, h, pts2, n, 1, 30, 0.01, true, rp.data(), rs.data());
        for (int i = 0; i < n; ++i) {
            float dx = pts1[i].x - rp[i].x, dy = pts1[i].y - rp[i].y;
            status[i] = (status[i] && rs[i] && std::sqrt(dx * dx + dy * dy) <= dist_thresh) ? 1 : 0;
        }
    }
    for (int i = 0; i < n; ++i)
        if (status[i] && !in_border(pts2[i], h, w)) status[i] = 0;
}

// cornerMinEigenVal(img, eig, blockSize=3, ksize=3), BORDER_DEFAULT (corner.cpp)
void min_eigen(const uint8_t* img, int w, int h, float* eig) {
    const double scale_d = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)(1.0 * scale_d), k2 = (float)(2.0 * scale_d);     // smoothing kernel [1 2 1]*scale as CV_32F
    std::vector<float> xx((size_t)w * h), xy((size_t)w * h), yy((size_t)w * h);
    parallel_rows(h, 32, [&](int ya, int yb) {
    for (int y = ya; y < yb; ++y) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * w;
        const uint8_t* r1 = img + (size_t)y * w;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x) {
            int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            // Dx: row filter [-1 0 1] (exact), column filter (S0+S2)*k1 + S1*k2
            float d0 = (float)(r0[xr] - r0[xl]), d1 = (float)(r1[xr] - r1[xl]), d2 = (float)(r2[xr] - r2[xl]);
            float dx = (d0 + d2) * k1 + d1 * k2;
            // Dy: row filter ((k1*a + k2*b) + k1*c) per row, column filter S2 - S0
            float s0 = (k1 * (float)r0[xl] + k2 * (float)r0[x]) + k1 * (float)r0[xr];
            float s2 = (k1 * (float)r2[xl] + k2 * (float)r2[x]) + k1 * (float)r2[xr];
            float dy = s2 - s0;
            size_t i = (size_t)y * w + x;
            xx[i] = dx * dx; xy[i] = dx * dy; yy[i] = dy * dy;
        }
    }
    });
    const bool box_f32 = g_var_box_sums == 1;
    auto box = [&](const std::vector<float>& c, int x, int y) -> float {      // D2
        if (box_f32) {
            float fs[3];
            for (int k = 0; k < 3; ++k) { const float* r = c.data() + (size_t)reflect101(y + k - 1, h) * w; fs[k] = (r[reflect101(x - 1, w)] + r[x]) + r[reflect101(x + 1, w)]; }
            return (fs[0] + fs[1]) + fs[2];
        }
        double rs[3];
        for (int k = 0; k < 3; ++k) {
            const float* r = c.data() + (size_t)reflect101(y + k - 1, h) * w;
            rs[k] = ((double)r[reflect101(x - 1, w)] + (double)r[x]) + (double)r[reflect101(x + 1, w)];
        }
        return (float)((rs[0] + rs[1]) + rs[2]);
    };
    parallel_rows(h, 32, [&](int ya, int yb) {
    for (int y = ya; y < yb; ++y)
        for (int x = 0; x < w; ++x) {
            float a = box(xx, x, y) * 0.5f, b = box(xy, x, y), c = box(yy, x, y) * 0.5f;
            eig[(size_t)y * w + x] = (float)((a + c) - std::sqrt((a - c) * (a - c) + b * b));
        }
    });
}

void gftt(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n, double quality, double min_dist,
          std::vector<P2f>& out) {           // featureselect.cpp goodFeaturesToTrack
    out.clear();
    std::vector<float> eig((size_t)w * h);
    min_eigen(img, w, h, eig.data());
    double maxVal = 0; bool any = false;
    for (size_t i = 0; i < eig.size(); ++i)
        if (!mask || mask[i]) { if (!any || eig[i] > maxVal) { maxVal = eig[i]; any = true; } }
    if (!any) maxVal = 0;
    const float thr = (float)(maxVal * quality);
    for (auto& v : eig) v = v > thr ? v : 0.f;           // THRESH_TOZERO
    std::vector<int> cand;
    {
        std::vector<std::vector<int>> rows((size_t)std::max(h, 1));      // per-row lists, concatenated in raster order: the same list whatever the thread count
        parallel_rows(std::max(h - 2, 0), 32, [&](int ya, int yb) {
            for (int y = ya + 1; y < yb + 1; ++y)
                for (int x = 1; x < w - 1; ++x) {
                    float v = eig[(size_t)y * w + x];
                    if (v == 0 || (mask && !mask[(size_t)y * w + x])) continue;
                    float m = v;
                    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) m = std::max(m, eig[(size_t)(y + dy) * w + x + dx]);
                    if (v == m) rows[y].push_back(y * w + x);
                }
        });
        for (auto& r : rows) cand.insert(cand.end(), r.begin(), r.end());
    }
    if (cand.empty()) return;
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {      // greaterThanPtr: value desc, then address desc
        return eig[a] > eig[b] ? true : eig[a] < eig[b] ? false : a > b; });
    if (min_dist >= 1) {
        const int cell = cv_round(min_dist);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<P2f>> grid((size_t)gw * gh);
        const float md2 = (float)(min_dist * min_dist);
        for (int ofs : cand) {
            int y = ofs / w, x = ofs - y * w;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (const P2f& m : grid[(size_t)yy * gw + xx]) {
                        float dx = x - m.x, dy = y - m.y;
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
            if (good) {
                grid[(size_t)yc * gw + xc].push_back({ (float)x, (float)y });
                out.push_back({ (float)x, (float)y });
                if (max_n > 0 && (int)out.size() == max_n) break;
            }
        }
    } else {
        for (int ofs : cand) {
            int y = ofs / w, x = ofs - y * w;
            out.push_back({ (float)x, (float)y });
            if (max_n > 0 && (int)out.size() == max_n) break;
        }
    }
}

// drawing.cpp Circle(img, center, radius, color, fill=true): midpoint circle, filled by hlines
void circle_zero(uint8_t* mask, int w, int h, int cx, int cy, int radius) {
    auto hline = [&](int y, int xa, int xb) {
        if ((unsigned)y >= (unsigned)h) return;
        xa = std::max(xa, 0); xb = std::min(xb, w - 1);
        for (int x = xa; x <= xb; ++x) mask[(size_t)y * w + x] = 0;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(cy - dy, cx - dx, cx + dx); hline(cy + dy, cx - dx, cx + dx);
        hline(cy - dx, cx - dy, cx + dy); hline(cy + dx, cx - dy, cx + dy);
        dy++; err += plus; plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
}

void lift_projective(const dvo_cam& c, double px, double py, double& ox, double& oy) {   // PinholeCamera.cc:450-508
    const double ik11 = 1.0 / c.fx, ik13 = -c.cx / c.fx, ik22 = 1.0 / c.fy, ik23 = -c.cy / c.fy;
    double mx_d = ik11 * px + ik13, my_d = ik22 * py + ik23;
    auto dist = [&](double ux, double uy, double& dxo, double& dyo) {     // :646-662
        double mx2 = ux * ux, my2 = uy * uy, mxy = ux * uy, rho2 = mx2 + my2;
        double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
        dxo = ux * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
        dyo = uy * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
    };
    bool no_dist = (c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0);   // m_noDistortion
    double mx_u, my_u;
    if (no_dist) { mx_u = mx_d; my_u = my_d; }
    else {
        double ddx, ddy;
        dist(mx_d, my_d, ddx, ddy);
        mx_u = mx_d - ddx; my_u = my_d - ddy;
        for (int i = 1; i < 8; ++i) { dist(mx_u, my_u, ddx, ddy); mx_u = mx_d - ddx; my_u = my_d - ddy; }
    }
    ox = mx_u / 1.0; oy = my_u / 1.0;    // b.x()/b.z() with z = 1
}

template <class T> void reduce_vector(std::vector<T>& v, const std::vector<uint8_t>& st) {   // feature_utils.h:77-85
    size_t j = 0;
    for (size_t i = 0; i < v.size(); ++i) if (st[i]) v[j++] = v[i];
    v.resize(j);
}

} // namespace

extern "C" void dvo_erode(const uint8_t* src, int w, int h, int k, uint8_t* dst);
struct dvo_tracker {
    dvo_fe_config cfg;
    Img prev0;
    double prev_time = 0, cur_time = 0;
    uint32_t global_id_count = 1;                 // InstFeat::global_id_count (instance_feature.h:137)
    std::vector<uint32_t> ids, right_ids;
    std::vector<int> track_cnt;
    std::vector<P2f> curr_points, curr_un, last_points, right_points, right_un, vel, right_vel;
    std::map<uint32_t, P2f> prev_id_pts, curr_id_pts, right_prev_id_pts, right_curr_id_pts;

    void undistort(const dvo_cam& cam, const std::vector<P2f>& in, std::vector<P2f>& out) {   // instance_feature.cpp:94-103
        out.clear();
        for (auto& p : in) { double x, y; lift_projective(cam, p.x, p.y, x, y); out.push_back({ (float)x, (float)y }); }
    }
    static void pts_velocity(double dt, const std::vector<uint32_t>& id, const std::vector<P2f>& un,
                             std::map<uint32_t, P2f>& cur_map, const std::map<uint32_t, P2f>& prev_map, std::vector<P2f>& v) {
        v.clear(); cur_map.clear();                                                              // instance_feature.cpp:26-85
        for (size_t i = 0; i < id.size(); ++i) cur_map.insert({ id[i], un[i] });
        for (size_t i = 0; i < un.size(); ++i) {
            auto it = prev_map.find(id[i]);
            if (!prev_map.empty() && it != prev_map.end()) {
                double vx = (un[i].x - it->second.x) / dt, vy = (un[i].y - it->second.y) / dt;
                v.push_back({ (float)vx, (float)vy });
            } else v.push_back({ 0.f, 0.f });
        }
    }
    int pack(dvo_feat* out) {                                                                    // background_tracker.cpp:340-392
        std::map<uint32_t, size_t> ridx;
        for (size_t i = 0; i < right_ids.size(); ++i) ridx[right_ids[i]] = i;
        for (size_t i = 0; i < ids.size(); ++i) {
            dvo_feat& f = out[i];
            std::memset(&f, 0, sizeof(f));
            f.id = ids[i]; f.track_cnt = track_cnt[i];
            double l[7] = { curr_un[i].x, curr_un[i].y, 1, curr_points[i].x, curr_points[i].y, vel[i].x, vel[i].y };
            std::memcpy(f.left, l, sizeof(l));
            auto it = ridx.find(ids[i]);
            if (cfg.stereo && it != ridx.end()) {
                size_t k = it->second; f.has_right = 1;
                double r[7] = { right_un[k].x, right_un[k].y, 1, right_points[k].x, right_points[k].y, right_vel[k].x, right_vel[k].y };
                std::memcpy(f.right, r, sizeof(r));
            }
        }
        return (int)ids.size();
    }
    // mode 0 TrackImage, 1 TrackImageNaive, 2 TrackSemanticImage (background_tracker.cpp:52-158, 400-516, 757-837)
    int track(const uint8_t* g0, const uint8_t* g1, const uint8_t* in_mask_raw, int mode, int erode_k, double time, dvo_feat* out) {
        const int w = cfg.width, h = cfg.height;
        const bool naive = mode != 0;
        cur_time = time;
        std::vector<uint8_t> mask((size_t)w * h, 255), eroded;
        const uint8_t* in_mask = in_mask_raw;
        if (naive && in_mask_raw && erode_k > 0) {               // use_mask_morphology (background_tracker.cpp:408-416,764-768)
            eroded.resize((size_t)w * h);
            dvo_erode(in_mask_raw, w, h, erode_k, eroded.data());
            in_mask = eroded.data();
        }
        if (in_mask) std::memcpy(mask.data(), in_mask, mask.size());
        const float dthr = mode == 1 ? 1.0f : 0.5f;                 // Q12: FeatureTrackByLK <= 0.5 (feature_utils.cpp:56), ...ByLKGpu <= 1.0 (:126)
        const float dthr_right = mode == 0 ? 0.5f : 1.0f;           // TrackRightGPU in naive and semantic mode
        curr_points.clear();
        if (!last_points.empty()) {
            curr_points.resize(last_points.size());
            std::vector<uint8_t> st(last_points.size());
            if (mode == 1 && !dvo_get_variant("f4_cpu_rule"))       // TrackLeftGPU -> FeatureTrackByLKGpu: the GPU tracker (lk_cuda.cpp); variant "f4_cpu_rule": the CPU arithmetic with the 1.0 px threshold (rounds 1-3)
                dvo_track_by_lk_gpu(prev0.d.data(), g0, w, h, &last_points[0].x, (int)last_points.size(), cfg.flow_back, &curr_points[0].x, st.data());
            else
                track_by_lk(prev0.d.data(), g0, w, h, last_points.data(), (int)last_points.size(), cfg.flow_back, dthr, curr_points.data(), st.data());
            if (naive && in_mask)                                  // instance_feature.cpp:211-216 (mask.at<uchar>(Point2f) rounds)
                for (size_t i = 0; i < st.size(); ++i)
                    if (st[i]) { int x = cv_round(curr_points[i].x), y = cv_round(curr_points[i].y); if (in_mask[(size_t)y * w + x] == 0) st[i] = 0; }
            reduce_vector(last_points, st); reduce_vector(curr_points, st); reduce_vector(ids, st); reduce_vector(track_cnt, st);
        }
        for (auto& c : track_cnt) c++;
        if (!naive) {                                               // SortPoints (D3: stable)
            std::vector<size_t> ord(ids.size());
            for (size_t i = 0; i < ord.size(); ++i) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return track_cnt[a] > track_cnt[b]; });
            std::vector<P2f> cp; std::vector<uint32_t> id2; std::vector<int> tc;
            for (size_t k : ord) { cp.push_back(curr_points[k]); id2.push_back(ids[k]); tc.push_back(track_cnt[k]); }
            curr_points.swap(cp); ids.swap(id2); track_cnt.swap(tc);
        }
        int n_max = cfg.max_cnt - (int)curr_points.size();
        bool detect = naive ? (n_max >= 10) : (n_max > 0);          // Q23
        if (detect) {
            for (auto& p : curr_points) circle_zero(mask.data(), w, h, cv_round(p.x), cv_round(p.y), cfg.min_dist);
            std::vector<P2f> npts;
            if (mode == 1 && !g_var_f5_cpu) {                       // DetectNewFeature(img, use_gpu = true, ...) (background_tracker.cpp:445): the GPU detector (gftt_cuda.cpp); variant "f5_cpu_rule": the CPU detector (rounds 1-5)
                npts.resize((size_t)std::max(n_max, 0)); int nn = 0;
                dvo_gftt_cuda(g0, mask.data(), w, h, n_max, 0.01, cfg.min_dist, npts.empty() ? nullptr : &npts[0].x, &nn);
                npts.resize((size_t)nn);
            } else
                gftt(g0, mask.data(), w, h, n_max, 0.01, cfg.min_dist, npts);
            for (auto& p : npts) { curr_points.push_back(p); ids.push_back(global_id_count++); track_cnt.push_back(1); }
        }
        undistort(cfg.cam0, curr_points, curr_un);
        pts_velocity(cur_time - prev_time, ids, curr_un, curr_id_pts, prev_id_pts, vel);
        if (cfg.stereo && g1) {
            right_ids.clear(); right_points.clear(); right_un.clear(); right_vel.clear(); right_curr_id_pts.clear();
            if (!curr_points.empty()) {
                right_points.resize(curr_points.size());
                std::vector<uint8_t> st(curr_points.size());
                if (mode != 0 && !dvo_get_variant("f4_cpu_rule"))   // TrackRightGPU -> FeatureTrackByLKGpu in naive and semantic mode
                    dvo_track_by_lk_gpu(g0, g1, w, h, &curr_points[0].x, (int)curr_points.size(), cfg.flow_back, &right_points[0].x, st.data());
                else
                    track_by_lk(g0, g1, w, h, curr_points.data(), (int)curr_points.size(), cfg.flow_back, dthr_right, right_points.data(), st.data());
                right_ids = ids;
                reduce_vector(right_points, st); reduce_vector(right_ids, st);
                undistort(cfg.cam1, right_points, right_un);
                pts_velocity(cur_time - prev_time, right_ids, right_un, right_curr_id_pts, right_prev_id_pts, right_vel);
            }
            right_prev_id_pts = right_curr_id_pts;
        }
        prev0.w = w; prev0.h = h; prev0.d.assign(g0, g0 + (size_t)w * h);
        prev_time = cur_time;
        last_points = curr_points;             // PostProcess (instance_feature.h:88-101)
        prev_id_pts = curr_id_pts;
        right_prev_id_pts = right_curr_id_pts;
        return pack(out);
    }
};

extern "C" {

void dvo_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) { pyr_down(src, w, h, dst); }
void dvo_scharr(const uint8_t* src, int w, int h, int16_t* out) { scharr(src, w, h, out); }
void dvo_lk(const uint8_t* a, const uint8_t* b, int w, int h, const float* pa, int n, int max_level, int iters, double eps,
            int use_initial, float* pb, uint8_t* status) {
    lk(a, b, w, h, (const P2f*)pa, n, max_level, iters, eps, use_initial != 0, (P2f*)pb, status);
}
void dvo_track_by_lk(const uint8_t* i1, const uint8_t* i2, int w, int h, const float* p1, int n, int flow_back, float dist_thresh,
                     float* p2, uint8_t* status) {
    track_by_lk(i1, i2, w, h, (const P2f*)p1, n, flow_back != 0, dist_thresh, (P2f*)p2, status);
}
void dvo_min_eigen(const uint8_t* img, int w, int h, float* eig) { min_eigen(img, w, h, eig); }
void dvo_gftt(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n, double quality, double min_dist, float* out_xy, int* n_out) {
    std::vector<P2f> out; gftt(img, mask, w, h, max_n, quality, min_dist, out);
    for (size_t i = 0; i < out.size(); ++i) { out_xy[2 * i] = out[i].x; out_xy[2 * i + 1] = out[i].y; }
    *n_out = (int)out.size();
}
void dvo_circle_mask(uint8_t* mask, int w, int h, const float* pts, int n, int radius) {
    for (int i = 0; i < n; ++i) circle_zero(mask, w, h, cv_round(pts[2 * i]), cv_round(pts[2 * i + 1]), radius);
}
void dvo_erode(const uint8_t* src, int w, int h, int k, uint8_t* dst) {
    // min over the k x k rectangle, anchor (-1,-1) -> centre = k/2, border = +inf for erode.  The minimum over a rectangle is the minimum over its rows of the
    // row minima: two 1-D passes give exactly the 2-D result (k = 20 on 1280x720 would be 3.7e8 comparisons per frame in the direct form).
    const int a = k / 2;
    std::vector<uint8_t> tmp((size_t)w * h);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        int m = 255;
        for (int dx = -a; dx < k - a; ++dx) { const int xx = x + dx; if (xx >= 0 && xx < w) m = std::min(m, (int)src[(size_t)y * w + xx]); }
        tmp[(size_t)y * w + x] = (uint8_t)m;
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        int m = 255;
        for (int dy = -a; dy < k - a; ++dy) { const int yy = y + dy; if (yy >= 0 && yy < h) m = std::min(m, (int)tmp[(size_t)yy * w + x]); }
        dst[(size_t)y * w + x] = (uint8_t)m;
    }
}
void dvo_lift_projective(const dvo_cam* cam, const float* pts, int n, float* out) {
    for (int i = 0; i < n; ++i) { double x, y; lift_projective(*cam, pts[2 * i], pts[2 * i + 1], x, y); out[2 * i] = (float)x; out[2 * i + 1] = (float)y; }
}
// VIODE::SetViodeMaskSimple / BuildViodeMask (utils/dataset/viode_utils.cpp:21-170): key = r*1000000 + g*1000*b (viode_utils.h:23-26, sic)
void dvo_viode_mask(const uint8_t* seg, int w, int h, int stride, const uint32_t* dyn_keys, int nkeys, uint8_t* merge, uint8_t* inv, uint32_t* key_img, int32_t* boxes) {
    for (int k = 0; k < nkeys; ++k) { boxes[4 * k] = boxes[4 * k + 1] = boxes[4 * k + 2] = boxes[4 * k + 3] = -1; }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const uint8_t* p = seg + (size_t)y * stride + 3 * x;
        const uint32_t key = (uint32_t)p[2] * 1000000u + (uint32_t)p[1] * 1000u * (uint32_t)p[0];
        int hit = -1;
        for (int k = 0; k < nkeys; ++k) if (dyn_keys[k] == key) { hit = k; break; }
        merge[(size_t)y * w + x] = hit >= 0 ? 255 : 0; inv[(size_t)y * w + x] = hit >= 0 ? 0 : 255;
        if (key_img) key_img[(size_t)y * w + x] = key;
        if (hit >= 0) {
            int32_t* b = boxes + 4 * hit;
            if (b[1] < 0) { b[0] = b[1] = y; b[2] = b[3] = x; }
            else { b[0] = std::min(b[0], y); b[1] = std::max(b[1], y); b[2] = std::min(b[2], x); b[3] = std::max(b[3], x); }
        }
    }
}
// cv::cvtColor(BGR2GRAY), 8U: fixed-point weights of color_yuv / color_rgb (B2Y 1868, G2Y 9617, R2Y 4899, shift 14)
void dvo_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray) {
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const uint8_t* p = bgr + (size_t)y * stride + 3 * x;
        gray[(size_t)y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
    }
}
// cv::remap(src, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT, 0) on 8-bit images with 1 or 3 channels — the call of
// ImageProcessor::Run (image_process/image_process.cpp:109-121) and of SemanticImage::SetMask on the merged mask (basic/semantic_image.cpp:86-89).
// Restated from OpenCV 3.4 imgproc (un-vendored; PARITY UNPINNED): remapBilinear<FixedPtCast<int, uchar, 15>, RemapVec_8u, short> with the
// fixed-point table of initInterTab2D(INTER_LINEAR, fixpt): INTER_BITS 5, weights saturate_cast<short>(wy * wx * 32768) whose sum is repaired to
// 32768 exactly as the library does (only the (0,0) cell needs it: {32767, 0, 0, 1}); out = (sum w v + 2^14) >> 15; neighbours outside the
// source take the border value 0.  All integer arithmetic: the SIMD and scalar paths of the library agree.
static const short* remap_tab() {
    static short tab[32 * 32 * 4 + 8]; static bool init = false;
    if (!init) {
        float t1[32][2];
        for (int i = 0; i < 32; ++i) { const float x = (float)i * (1.f / 32); t1[i][0] = 1.f - x; t1[i][1] = x; }      // interpolateLinear
        short* it = tab;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j, it += 4) {
            int isum = 0;
            for (int k1 = 0; k1 < 2; ++k1) for (int k2 = 0; k2 < 2; ++k2) {
                const float v = t1[i][k1] * t1[j][k2];
                const int r = (int)std::lrint((double)(v * 32768.f));
                isum += it[k1 * 2 + k2] = (short)std::min(std::max(r, -32768), 32767);
            }
            if (isum != 32768) {          // the library scans it[k1 * 2 + k2], k1, k2 in {1, 2}: past this cell's 4 entries into still-zero memory
                const int diff = isum - 32768; int Mk = 3, mk = 3;
                for (int k1 = 1; k1 < 3; ++k1) for (int k2 = 1; k2 < 3; ++k2) { const int k = k1 * 2 + k2; if (it[k] < it[mk]) mk = k; else if (it[k] > it[Mk]) Mk = k; }
                if (diff < 0) it[Mk] = (short)(it[Mk] - diff); else it[mk] = (short)(it[mk] - diff);
            }
        }
        init = true;
    }
    return tab;
}
void dvo_remap(const uint8_t* src, int w, int h, int stride, int cn, const int16_t* map1, const uint16_t* map2, uint8_t* dst) {
    const short* wtab = remap_tab();
    for (int dy = 0; dy < h; ++dy) for (int dx = 0; dx < w; ++dx) {
        const int sx = map1[((size_t)dy * w + dx) * 2], sy = map1[((size_t)dy * w + dx) * 2 + 1];
        const short* wt = wtab + 4 * (map2[(size_t)dy * w + dx] & 1023);
        for (int c = 0; c < cn; ++c) {
            auto at = [&](int x, int y) -> int { return ((unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h) ? src[(size_t)y * stride + (size_t)x * cn + c] : 0; };
            const int v = at(sx, sy) * wt[0] + at(sx + 1, sy) * wt[1] + at(sx, sy + 1) * wt[2] + at(sx + 1, sy + 1) * wt[3];
            dst[((size_t)dy * w + dx) * cn + c] = (uint8_t)std::min(std::max((v + (1 << 14)) >> 15, 0), 255);
        }
    }
}
// cv::initUndistortRectifyMap(K, D, Mat(), newK, size, CV_16SC2, map1, map2) for the pinhole + radtan(k1, k2, p1, p2) cameras of the reference
// (utils/camera_model.cpp:481-499); used by the tests to produce realistic maps.  OpenCV 3.4 calib3d/imgproc undistort.cpp, restated.
void dvo_init_undistort_map(const dvo_cam* cam, const double* newK4 /* fx fy cx cy */, int w, int h, int16_t* map1, uint16_t* map2) {
    const double ir[9] = { 1.0 / newK4[0], 0, -newK4[2] / newK4[0], 0, 1.0 / newK4[1], -newK4[3] / newK4[1], 0, 0, 1 };      // (newK I)^-1
    const double k1 = cam->k1, k2 = cam->k2, p1 = cam->p1, p2 = cam->p2;
    for (int i = 0; i < h; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double iw = 1. / _w, x = _x * iw, y = _y * iw;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((0 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2), yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
            const double u = cam->fx * xd + cam->cx, v = cam->fy * yd + cam->cy;
            const int iu = (int)std::lrint(u * 32), iv = (int)std::lrint(v * 32);
            map1[((size_t)i * w + j) * 2] = (int16_t)(iu >> 5); map1[((size_t)i * w + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(size_t)i * w + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}
dvo_tracker* dvo_tracker_create(const dvo_fe_config* cfg) { auto* t = new dvo_tracker(); t->cfg = *cfg; return t; }
void dvo_tracker_destroy(dvo_tracker* t) { delete t; }
int dvo_tracker_track_image(dvo_tracker* t, const uint8_t* g0, const uint8_t* g1, double time, dvo_feat* out) {
    return t->track(g0, g1, nullptr, 0, 0, time, out);
}
int dvo_tracker_track_image_naive(dvo_tracker* t, const uint8_t* g0, const uint8_t* g1, const uint8_t* mask, double time, dvo_feat* out) {
    return t->track(g0, g1, mask, 1, 0, time, out);
}
int dvo_tracker_track_image_mode(dvo_tracker* t, const uint8_t* g0, const uint8_t* g1, const uint8_t* mask, int mode, int erode_k, double time, dvo_feat* out) {
    return t->track(g0, g1, mask, mode, erode_k, time, out);
}

// One visible object instance through one frame of InstsFeatManager::InstsTrack (front_end/dynamic_tracker.cpp:348-470):
//   InstanceImagePadding (feature_utils.cpp:406-413) -> InstFeat::TrackLeft without mask (instance_feature.cpp:149-188)
//   -> ErodeMask 5x5 + cv::circle(kMinDynamicDist) + goodFeaturesToTrack(roi_gray, kMaxDynamicCnt - n, 0.01, kMinDynamicDist)
//   -> UndistortedPointsWithAddOffset (:123-133) -> TrackRightByPad on the full images (:251-275) -> RightUndistortedPts.
// Points live in ROI coordinates (previous ROI for last_pts, current ROI for the outputs), as in the reference.
// The VIODE segmentation-key test of TrackRightByPad belongs to SURVEY row N4 and is not applied.
int dvo_inst_track(const uint8_t* prev_roi, int pw, int ph, const uint8_t* cur_roi, int cw, int ch, const uint8_t* cur_mask, int box_x, int box_y,
                   const uint8_t* gray0, const uint8_t* gray1, int W, int H, const dvo_cam* cam0, const dvo_cam* cam1,
                   int n_last, const float* last_pts, const uint32_t* ids, const int32_t* track_cnt,
                   int max_cnt, int min_dist, int flow_back, uint32_t* global_id,
                   int* n_cur, float* cur_pts, uint32_t* cur_ids, int32_t* cur_cnt, float* cur_un,
                   int* n_right, float* right_pts, uint32_t* right_ids, float* right_un) {
    std::vector<P2f> last, cur; std::vector<uint32_t> id; std::vector<int> cnt;
    if (prev_roi && n_last > 0) {
        const int w = std::max(pw, cw), h = std::max(ph, ch);
        std::vector<uint8_t> a((size_t)w * h, 0), b((size_t)w * h, 0);
        for (int y = 0; y < ph; ++y) std::memcpy(&a[(size_t)y * w], prev_roi + (size_t)y * pw, pw);
        for (int y = 0; y < ch; ++y) std::memcpy(&b[(size_t)y * w], cur_roi + (size_t)y * cw, cw);
        last.resize(n_last); cur.resize(n_last);
        for (int i = 0; i < n_last; ++i) last[i] = { last_pts[2 * i], last_pts[2 * i + 1] };
        std::vector<uint8_t> st(n_last);
        track_by_lk(a.data(), b.data(), w, h, last.data(), n_last, flow_back != 0, 0.5f, cur.data(), st.data());
        id.assign(ids, ids + n_last); cnt.assign(track_cnt, track_cnt + n_last);
        reduce_vector(cur, st); reduce_vector(id, st); reduce_vector(cnt, st);
        for (auto& c : cnt) c++;
    }
    if ((int)cur.size() < max_cnt) {
        const int max_new = max_cnt - (int)cur.size();
        std::vector<uint8_t> m((size_t)cw * ch, 255);
        if (cur_mask) dvo_erode(cur_mask, cw, ch, 5, m.data());
        for (auto& p : cur) circle_zero(m.data(), cw, ch, cv_round(p.x), cv_round(p.y), min_dist);
        std::vector<P2f> npts;
        gftt(cur_roi, m.data(), cw, ch, max_new, 0.01, min_dist, npts);
        for (auto& p : npts) { cur.push_back(p); id.push_back((*global_id)++); cnt.push_back(1); }
    }
    *n_cur = (int)cur.size();
    for (size_t i = 0; i < cur.size(); ++i) {
        cur_pts[2 * i] = cur[i].x; cur_pts[2 * i + 1] = cur[i].y; cur_ids[i] = id[i]; cur_cnt[i] = cnt[i];
        double x, y; lift_projective(*cam0, (double)(cur[i].x + (float)box_x), (double)(cur[i].y + (float)box_y), x, y);      // float + float: Box2D::rect is a cv::Rect2f (basic/box2d.h:56)
        cur_un[2 * i] = (float)x; cur_un[2 * i + 1] = (float)y;
    }
    *n_right = 0;
    if (gray1 && !cur.empty()) {
        std::vector<P2f> padded(cur.size()), rp(cur.size());
        for (size_t i = 0; i < cur.size(); ++i) padded[i] = { cur[i].x + (float)box_x, cur[i].y + (float)box_y };
        std::vector<uint8_t> st(cur.size());
        track_by_lk(gray0, gray1, W, H, padded.data(), (int)padded.size(), flow_back != 0, 0.5f, rp.data(), st.data());
        std::vector<uint32_t> rid = id;
        reduce_vector(rp, st); reduce_vector(rid, st);
        *n_right = (int)rp.size();
        for (size_t i = 0; i < rp.size(); ++i) {
            right_pts[2 * i] = rp[i].x; right_pts[2 * i + 1] = rp[i].y; right_ids[i] = rid[i];
            double x, y; lift_projective(*cam1, rp[i].x, rp[i].y, x, y);
            right_un[2 * i] = (float)x; right_un[2 * i + 1] = (float)y;
        }
    }
    return 0;
}


// ------------------------------------------------------------------------------------------------------------------------------
// InstsFeatManager (front_end/dynamic_tracker.{h,cpp}) + InstFeat (front_end/instance_feature.{h,cpp}) with persistent state, as the
// reference's thread T2 drives them in dynamic mode (system/main.cpp:198-250): visibility reset -> AddInstancesByTracking (:791-828, the
// detections arrive with their track ids) -> InstsTrack (:348-493) -> Output (:521-577).  Canonical choices: objects are visited in ascending
// id (reference: unordered_map order), the object tracker runs after the background tracker and shares its id counter
// (InstFeat::global_id_count is one static, incremented from two threads in the reference).
struct OInstFeat {
    unsigned id = 0; int lost_num = 0; bool is_curr_visible = false, has_box2d = false, has_box3d = false;
    int class_id = 0, rx = 0, ry = 0, rw = 0, rh = 0;
    dvo_box3d box3d{};
    std::vector<uint32_t> ids, right_ids; std::vector<int> track_cnt;
    std::vector<P2f> curr_points, curr_un_points, last_points, right_points, right_un_points, pts_velocity, right_pts_velocity;
    std::map<uint32_t, P2f> prev_id_pts, curr_id_pts, right_prev_id_pts, right_curr_id_pts;
    Img roi_gray, prev_roi_gray, mask_cv;
    std::vector<double> extra_points3d;
};
struct dvo_insts {
    dvo_tracker* bg; int max_cnt, min_dist, use_det3d;
    std::map<unsigned, OInstFeat> instances;
    double curr_time = 0, last_time = 0;
    std::vector<float> disp; float baseline = 0; bool have_disp = false;      // curr_img.disp of the next frame (dvo_insts_set_disparity)
    std::vector<uint32_t> right_keys; bool have_keys = false;                 // PixelToKey of every pixel of curr_img.seg1 of the next frame (dvo_insts_set_right_keys): cfg::dataset == kViode
};
// VIODE: TrackRightByPad keeps a right-image point only if the segmentation image of the RIGHT camera carries the object's key there
// (instance_feature.cpp:263-268: status[i] && VIODE::PixelToKey(right_points[i], img.seg1) != id -> 0; Mat::at(Point2f) rounds like cvRound)
void dvo_insts_set_right_keys(dvo_insts* M, const uint32_t* key_img) {
    const size_t n = (size_t)M->bg->cfg.width * M->bg->cfg.height;
    M->have_keys = key_img != nullptr;
    if (key_img) M->right_keys.assign(key_img, key_img + n);
}
void dvo_insts_set_disparity(dvo_insts* M, const float* disp, float baseline) {
    const size_t n = (size_t)M->bg->cfg.width * M->bg->cfg.height;
    M->have_disp = disp != nullptr;
    if (disp) { M->disp.assign(disp, disp + n); M->baseline = baseline; }
}

static float o_rect_iou(const float a[4], const float b[4]) {      // Box2D::IoU (basic/box2d.cpp:20-27)
    const float x1 = std::max(a[0], b[0]), y1 = std::max(a[1], b[1]), x2 = std::min(a[0] + a[2], b[0] + b[2]), y2 = std::min(a[1] + a[3], b[1] + b[3]);
    const float in = (x2 > x1 && y2 > y1) ? (x2 - x1) * (y2 - y1) : 0.f;
    const float un = a[2] * a[3] + b[2] * b[3] - in;
    if (un < 2.220446049250313e-16) return 0.f;
    return in / un;
}

dvo_insts* dvo_insts_create(dvo_tracker* bg, int max_dynamic_cnt, int min_dynamic_dist, int use_det3d) { return new dvo_insts{ bg, max_dynamic_cnt, min_dynamic_dist, use_det3d }; }
void dvo_insts_destroy(dvo_insts* m) { delete m; }

int dvo_insts_track(dvo_insts* M, const uint8_t* gray0, const uint8_t* gray1, double time, const dvo_inst_det* dets, int n_dets, const dvo_box3d* boxes3d, int n_boxes3d) {
    const dvo_fe_config& cfg = M->bg->cfg;
    const int W = cfg.width, H = cfg.height;
    M->curr_time = time;
    for (auto& kv : M->instances) { kv.second.is_curr_visible = false; kv.second.has_box2d = false; kv.second.has_box3d = false; }      // main.cpp:198-202
    for (int i = 0; i < n_dets; ++i) {          // AddInstancesByTracking
        const dvo_inst_det& d = dets[i];
        OInstFeat& I = M->instances[d.track_id];
        I.id = d.track_id; I.is_curr_visible = true; I.has_box2d = true; I.class_id = d.class_id; I.rx = d.x; I.ry = d.y; I.rw = d.w; I.rh = d.h;
        I.roi_gray.w = d.w; I.roi_gray.h = d.h; I.roi_gray.d.resize((size_t)d.w * d.h);
        for (int y = 0; y < d.h; ++y) std::memcpy(&I.roi_gray.d[(size_t)y * d.w], gray0 + (size_t)(d.y + y) * W + d.x, d.w);      // roi_gray = gray0(rect) (basic/semantic_image.cpp:58-59)
        I.mask_cv.w = d.w; I.mask_cv.h = d.h; I.mask_cv.d.assign(d.mask, d.mask + (size_t)d.w * d.h);
        if (!M->have_disp) I.extra_points3d.assign(d.points ? d.points : nullptr, d.points ? d.points + 3 * (size_t)std::max(d.n_points, 0) : nullptr);      // pass-through form: the caller ran the extra-point pipeline
    }
    // ---- InstsTrack ----
    for (auto& kv : M->instances) { if (!kv.second.is_curr_visible) kv.second.lost_num++; else kv.second.lost_num = 0; }
    if (M->use_det3d) {          // BoxAssociate2Dto3D (:61-152)
        std::vector<char> match_vec(n_boxes3d, 0);
        for (auto& kv : M->instances) {
            OInstFeat& inst = kv.second;
            if (!inst.is_curr_visible) continue;
            const float inst_rect[4] = { (float)inst.rx, (float)inst.ry, (float)inst.rw, (float)inst.rh };
            double min_dist = 1.7976931348623157e308; int min_idx = -1;
            for (int i = 0; i < n_boxes3d; ++i) {
                if (match_vec[i]) continue;
                const dvo_box3d& b = boxes3d[i];
                const int x0 = cv_round(b.rect_min[0]), y0 = cv_round(b.rect_min[1]), x1 = cv_round(b.rect_max[0]), y1 = cv_round(b.rect_max[1]);
                const float proj_rect[4] = { (float)x0, (float)y0, (float)(x1 - x0), (float)(y1 - y0) };
                const float iou = o_rect_iou(inst_rect, proj_rect);
                if (inst.class_id != b.class_id) continue;
                if (iou > 0.1f) { const double n = std::sqrt(b.center[0] * b.center[0] + b.center[1] * b.center[1] + b.center[2] * b.center[2]); if (n < min_dist) { min_dist = n; min_idx = i; } }
            }
            if (min_idx >= 0) { match_vec[min_idx] = 1; inst.has_box3d = true; inst.box3d = boxes3d[min_idx]; }
        }
    }
    const bool is_exist_inst = n_dets > 0;
    auto exec = [&](auto f) { for (auto& kv : M->instances) { if (kv.second.lost_num > 0) continue; f(kv.second); } };
    if (is_exist_inst) {
        // ProcessExtraPoints (:159-340), the reference's second thread of this call: started before the optical flow, its sampling reads roi->mask_cv as the detection
        // delivered it (the in-place erosion below comes later on the other thread; see extra_points.cpp on the race)
        if (M->have_disp) exec([&](OInstFeat& inst) {
            if (!inst.is_curr_visible) return;
            std::vector<float> raw((size_t)3 * inst.mask_cv.w * inst.mask_cv.h), seg;
            const int n = dvo_detect_extra_points(inst.mask_cv.d.data(), inst.mask_cv.w, inst.mask_cv.h, inst.rx, inst.ry, M->disp.data(), W, H,
                                                  (float)cfg.cam0.fx, (float)cfg.cam0.fy, (float)cfg.cam0.cx, (float)cfg.cam0.cy, M->baseline, raw.data(), inst.mask_cv.w * inst.mask_cv.h);
            seg.resize((size_t)3 * std::max(n, 1));
            const int m = dvo_process_extra_points(raw.data(), n, seg.data());
            inst.extra_points3d.clear();                                       // PclToEigen: float -> double
            for (int k = 0; k < 3 * m; ++k) inst.extra_points3d.push_back((double)seg[k]);
        });
        M->have_disp = false;
        exec([&](OInstFeat& inst) {      // per-object optical flow (:381-413)
            if (!inst.is_curr_visible) return;
            if (inst.prev_roi_gray.d.empty() || inst.last_points.empty()) return;
            const int w = std::max(inst.prev_roi_gray.w, inst.roi_gray.w), h = std::max(inst.prev_roi_gray.h, inst.roi_gray.h);      // InstanceImagePadding
            std::vector<uint8_t> a((size_t)w * h, 0), b((size_t)w * h, 0);
            for (int y = 0; y < inst.prev_roi_gray.h; ++y) std::memcpy(&a[(size_t)y * w], &inst.prev_roi_gray.d[(size_t)y * inst.prev_roi_gray.w], inst.prev_roi_gray.w);
            for (int y = 0; y < inst.roi_gray.h; ++y) std::memcpy(&b[(size_t)y * w], &inst.roi_gray.d[(size_t)y * inst.roi_gray.w], inst.roi_gray.w);
            // InstFeat::TrackLeft(curr, last) (instance_feature.cpp:149-188)
            inst.curr_points.assign(inst.last_points.size(), P2f{ 0, 0 });
            std::vector<uint8_t> status(inst.last_points.size());
            track_by_lk(a.data(), b.data(), w, h, inst.last_points.data(), (int)inst.last_points.size(), cfg.flow_back != 0, 0.5f, inst.curr_points.data(), status.data());
            reduce_vector(inst.curr_points, status); reduce_vector(inst.ids, status); reduce_vector(inst.last_points, status); reduce_vector(inst.track_cnt, status);
            for (auto& n : inst.track_cnt) n++;
        });
        exec([&](OInstFeat& inst) {      // corner detection (:418-446)
            if ((int)inst.curr_points.size() >= M->max_cnt) return;
            const int max_new_detect = M->max_cnt - (int)inst.curr_points.size();
            std::vector<uint8_t> er((size_t)inst.mask_cv.w * inst.mask_cv.h);
            dvo_erode(inst.mask_cv.d.data(), inst.mask_cv.w, inst.mask_cv.h, 5, er.data());
            inst.mask_cv.d = er;                                               // ErodeMask(mask_cv, mask_cv, 5) works in place
            std::vector<uint8_t> inst_mask = inst.mask_cv.d;
            for (size_t i = 0; i < inst.curr_points.size(); ++i) circle_zero(inst_mask.data(), inst.mask_cv.w, inst.mask_cv.h, cv_round(inst.curr_points[i].x), cv_round(inst.curr_points[i].y), M->min_dist);
            std::vector<P2f> new_pts;
            gftt(inst.roi_gray.d.data(), inst_mask.data(), inst.roi_gray.w, inst.roi_gray.h, max_new_detect, 0.01, M->min_dist, new_pts);
            for (auto& pt : new_pts) { inst.curr_points.push_back(pt); inst.ids.push_back(M->bg->global_id_count++); inst.track_cnt.push_back(1); }
        });
        const double dt = M->curr_time - M->last_time;
        for (auto& kv : M->instances) {
            OInstFeat& inst = kv.second;
            if (!inst.is_curr_visible) continue;
            inst.curr_un_points.clear();                                     // UndistortedPointsWithAddOffset (instance_feature.cpp:123-133): float + float, then widened
            for (auto& pt : inst.curr_points) { double x, y; lift_projective(cfg.cam0, (double)(pt.x + (float)inst.rx), (double)(pt.y + (float)inst.ry), x, y); inst.curr_un_points.push_back({ (float)x, (float)y }); }
            dvo_tracker::pts_velocity(dt, inst.ids, inst.curr_un_points, inst.curr_id_pts, inst.prev_id_pts, inst.pts_velocity);
        }
        if (gray1 && cfg.stereo) exec([&](OInstFeat& inst) {      // TrackRightByPad + RightUndistortedPts + RightPtsVelocity (:462-471)
            if (!inst.is_curr_visible) return;
            if (!inst.curr_points.empty()) {
                inst.right_points.assign(inst.curr_points.size(), P2f{ 0, 0 });
                std::vector<P2f> padded(inst.curr_points.size());
                for (size_t i = 0; i < padded.size(); ++i) padded[i] = { inst.curr_points[i].x + (float)inst.rx, inst.curr_points[i].y + (float)inst.ry };
                std::vector<uint8_t> status(padded.size());
                track_by_lk(gray0, gray1, W, H, padded.data(), (int)padded.size(), cfg.flow_back != 0, 0.5f, inst.right_points.data(), status.data());
                if (M->have_keys)                                          // cfg::dataset == DatasetType::kViode (instance_feature.cpp:263-268)
                    for (size_t i = 0; i < status.size(); ++i)
                        if (status[i] && M->right_keys[(size_t)cv_round(inst.right_points[i].y) * W + cv_round(inst.right_points[i].x)] != inst.id) status[i] = 0;
                inst.right_ids = inst.ids;
                reduce_vector(inst.right_points, status); reduce_vector(inst.right_ids, status);
            }                                                              // (empty curr_points: TrackRightByPad returns early and right_points / right_ids keep their old content)
            inst.right_un_points.clear();
            for (auto& pt : inst.right_points) { double x, y; lift_projective(cfg.cam1, pt.x, pt.y, x, y); inst.right_un_points.push_back({ (float)x, (float)y }); }
            dvo_tracker::pts_velocity(dt, inst.right_ids, inst.right_un_points, inst.right_curr_id_pts, inst.right_prev_id_pts, inst.right_pts_velocity);
        });
    }
    // ManageInstances (:499-514)
    for (auto it = M->instances.begin(); it != M->instances.end();) {
        OInstFeat& inst = it->second;
        if (inst.lost_num == 0 && !inst.has_box2d) inst.lost_num++;
        bool erase = false;
        if (inst.lost_num > 0) { inst.lost_num++; if (inst.lost_num > 3) erase = true; }
        if (erase) it = M->instances.erase(it); else ++it;
    }
    M->have_keys = false;
    if (is_exist_inst) exec([&](OInstFeat& inst) {      // PostProcess (instance_feature.h:88-101)
        inst.last_points = inst.curr_points;
        inst.prev_id_pts = inst.curr_id_pts; inst.right_prev_id_pts = inst.right_curr_id_pts;
        inst.prev_roi_gray = inst.roi_gray;
    });
    M->last_time = M->curr_time;
    return 0;
}

// InstsFeatManager::Output (:521-577)
int dvo_insts_output(dvo_insts* M, dvo_inst_obs* insts, int cap_insts, int* n_insts, dvo_feat* feats, int cap_feats, int* n_feats, double* points, int cap_points, int* n_points) {
    int ki = 0, kf = 0, kp = 0;
    for (auto& kv : M->instances) {
        OInstFeat& inst = kv.second;
        if (inst.lost_num > 0 || !inst.is_curr_visible) continue;
        const int n = (int)inst.curr_un_points.size(), np = (int)(inst.extra_points3d.size() / 3);
        if (ki >= cap_insts || kf + n > cap_feats || kp + np > cap_points) return -1;
        dvo_inst_obs& o = insts[ki++];
        std::memset(&o, 0, sizeof(o));
        o.id = inst.id; o.has_box3d = inst.has_box3d; o.first_feat = kf; o.n_feats = n; o.first_point = kp; o.n_points = np;
        o.rect[0] = (float)inst.rx; o.rect[1] = (float)inst.ry; o.rect[2] = (float)inst.rw; o.rect[3] = (float)inst.rh;
        if (inst.has_box3d) o.box3d = inst.box3d;
        std::map<uint32_t, size_t> ridx;
        for (size_t i = 0; i < inst.right_ids.size() && i < inst.right_un_points.size(); ++i) ridx[inst.right_ids[i]] = i;
        for (int i = 0; i < n; ++i) {
            dvo_feat& f = feats[kf + i];
            std::memset(&f, 0, sizeof(f));
            f.id = inst.ids[i]; f.track_cnt = inst.track_cnt[i];
            const double l[7] = { inst.curr_un_points[i].x, inst.curr_un_points[i].y, 1, inst.curr_points[i].x, inst.curr_points[i].y, inst.pts_velocity[i].x, inst.pts_velocity[i].y };
            std::memcpy(f.left, l, sizeof(l));
            auto it = ridx.find(inst.ids[i]);
            if (M->bg->cfg.stereo && it != ridx.end()) {
                const size_t k = it->second; f.has_right = 1;
                const double r[7] = { inst.right_un_points[k].x, inst.right_un_points[k].y, 1, inst.right_points[k].x, inst.right_points[k].y, inst.right_pts_velocity[k].x, inst.right_pts_velocity[k].y };
                std::memcpy(f.right, r, sizeof(r));
            }
        }
        if (np) std::memcpy(points + 3 * (size_t)kp, inst.extra_points3d.data(), 24 * (size_t)np);
        kf += n; kp += np;
    }
    *n_insts = ki; *n_feats = kf; *n_points = kp;
    return 0;
}

} // extern "C"
