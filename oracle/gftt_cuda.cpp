// gftt_cuda.cpp — CPU ORACLE (test infrastructure, NOT the product) of the reference's GPU corner detector (second half):
//   DetectShiTomasiCornersGpu                    front_end/feature_utils.cpp:339-348     cv::cuda::createGoodFeaturesToTrackDetector(CV_8UC1, n, 0.01, min_dist)->detect(img, pts, mask)
// called with use_gpu = true by TrackImageNaive only (front_end/background_tracker.cpp:445 -> InstFeat::DetectNewFeature, front_end/instance_feature.cpp:372-379);
// TrackImage, TrackSemanticImage and the per-object trackers call cv::goodFeaturesToTrack (front_oracle.cpp gftt()).
//
// OpenCV 3.4.16's cudaimgproc / cudafilters modules are NOT under /root/reference and the reference holds no vectors for them: restated from the published sources
// (modules/cudaimgproc/src/gftt.cpp, src/cuda/gftt.cu, src/corners.cpp, src/cuda/corners.cu; modules/cudafilters/src/filtering.cpp, src/cuda/row_filter.hpp,
// column_filter.hpp) — PARITY UNPINNED.  Where the GPU detector differs from cv::goodFeaturesToTrack (each difference decides WHICH corners are picked):
//   (1) the response map.  Both compute Sobel derivatives scaled by 1 / (4 * 3 * 255) and the smaller eigenvalue of the 3 x 3 block sums of (dx^2, dx dy, dy^2), but
//       the GPU code runs the separable filter as a float multiply-add chain over the taps in order (row filter, then column filter, the scale folded into the
//       smoothing taps [1 2 1]), sums the nine products of a block in raster order in float INSIDE the eigenvalue kernel (no covariance image, no boxFilter), and
//       evaluates (a + c) - sqrtf((a - c)^2 + b^2) in float; borders are BORDER_REFLECT101 for the filters and for the block.
//   (2) the threshold.  cuda::minMax(eig) is taken over the WHOLE image — the CPU code passes the mask to minMaxLoc.  With a mask (objects + the discs stamped around
//       the tracked points) the strongest response usually sits under a disc, so the GPU threshold is the higher one.
//   (3) the candidates.  findCorners: 0 < x < w-1, 0 < y < h-1, mask != 0, eig > threshold, eig == max of the raw 3 x 3 neighbourhood (the CPU code thresholds to zero
//       first and then compares with the dilated map; the accepted set is the same given the same map and threshold).
//   (4) the order.  thrust::sort by value, descending; then the same greedy minimum-distance grid as the CPU code, on the host, on float coordinates.
// Three things the library leaves to the platform are FIXED here by declaration (tests/test_oracle_variants.py bounds what each is worth):
//   (a) floating-point contraction.  nvcc's default is -fmad=true: x * y + z compiles to one FFMA.  The accumulations here are written `sum = sum + v * k` and
//       `a += dx * dx`, which contract; (a - c) * (a - c) + b * b contracts its FIRST product into the add (LLVM's fadd combine takes the left operand).  Canonical:
//       contracted (variant "gftt_cuda_fma" 0); variant 1: no contraction anywhere.
//   (b) ties.  Candidates are appended through an atomic counter (order = scheduling) and thrust::sort with a comparator is a stable merge sort: corners of EQUAL
//       response come out in scheduling order.  Canonical: address descending, the order of the CPU detector's comparator ("gftt_cuda_tie" 0); variant 1: ascending.
//   (c) the candidate buffer.  findCorners keeps the first max(1000, 0.05 * w * h) candidates the atomic counter hands out; beyond that the kept SET depends on the
//       schedule.  Here every candidate is kept (a frame would need more than 5 % of its pixels to be above-threshold local maxima).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include "dvo.h"

namespace {

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
inline int cv_round(double v) { return (int)std::nearbyint(v); }

// one tap of a filter chain: sum + v * k, fused or not
inline float mad(float v, float k, float sum, bool fused) { return fused ? std::fmaf(v, k, sum) : sum + v * k; }

// CornerBase::extractCovData (cudaimgproc/src/corners.cpp): filterDx_ = createSobelFilter(CV_8UC1, CV_32F, 1, 0, 3, scale), filterDy_ = (.., 0, 1, 3, scale),
// scale = 1 / ((1 << 2) * blockSize * 255); createSobelFilter scales the SMOOTHING kernel (kx for dy, ky for dx); SeparableLinearFilter: row pass u8 -> float
// buffer, column pass float -> float, `sum = sum + src * kernel[k]`, k = 0, 1, 2 (row_filter.hpp, column_filter.hpp)
void sobel_cuda(const uint8_t* img, int w, int h, bool fused, std::vector<float>& Dx, std::vector<float>& Dy) {
    const double scale = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float ks[3] = { (float)(1.0 * scale), (float)(2.0 * scale), (float)(1.0 * scale) };      // Mat *= double on CV_32F: computed in double, stored as float
    const float kd[3] = { -1.f, 0.f, 1.f };
    std::vector<float> bx((size_t)w * h), by((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float sx = 0.f, sy = 0.f;
            for (int k = 0; k < 3; ++k) {
                const float v = (float)img[(size_t)y * w + reflect101(x + k - 1, w)];
                sx = mad(v, kd[k], sx, fused);       // Dx: row kernel [-1 0 1]  (exact either way)
                sy = mad(v, ks[k], sy, fused);       // Dy: row kernel [1 2 1] * scale
            }
            bx[(size_t)y * w + x] = sx; by[(size_t)y * w + x] = sy;
        }
    Dx.resize((size_t)w * h); Dy.resize((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float sx = 0.f, sy = 0.f;
            for (int k = 0; k < 3; ++k) {
                const size_t r = (size_t)reflect101(y + k - 1, h) * w + x;
                sx = mad(bx[r], ks[k], sx, fused);   // Dx: column kernel [1 2 1] * scale
                sy = mad(by[r], kd[k], sy, fused);   // Dy: column kernel [-1 0 1]
            }
            Dx[(size_t)y * w + x] = sx; Dy[(size_t)y * w + x] = sy;
        }
}

// cornerMinEigenVal_kernel (cudaimgproc/src/cuda/corners.cu): the block sums of one pixel in raster order, then the closed form
void min_eigen_cuda(const uint8_t* img, int w, int h, float* eig) {
    const bool fused = dvo_get_variant("gftt_cuda_fma") == 0;
    std::vector<float> Dx, Dy;
    sobel_cuda(img, w, h, fused, Dx, Dy);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float a = 0.f, b = 0.f, c = 0.f;
            for (int i = y - 1; i <= y + 1; ++i) {
                const int yy = reflect101(i, h);
                for (int j = x - 1; j <= x + 1; ++j) {
                    const size_t q = (size_t)yy * w + reflect101(j, w);
                    const float dx = Dx[q], dy = Dy[q];
                    a = mad(dx, dx, a, fused); b = mad(dx, dy, b, fused); c = mad(dy, dy, c, fused);
                }
            }
            a *= 0.5f; c *= 0.5f;
            const float d = a - c;
            const float rad = fused ? std::fmaf(d, d, b * b) : d * d + b * b;
            eig[(size_t)y * w + x] = (a + c) - std::sqrt(rad);        // sqrtf: correctly rounded (-prec-sqrt=true is nvcc's default)
        }
}

struct P2 { float x, y; };

// GoodFeaturesToTrackDetector::detect (cudaimgproc/src/gftt.cpp) with findCorners / sortCorners (src/cuda/gftt.cu)
void gftt_cuda(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n, double quality, double min_dist, std::vector<P2>& out) {
    out.clear();
    if (w <= 0 || h <= 0) return;
    std::vector<float> eig((size_t)w * h);
    min_eigen_cuda(img, w, h, eig.data());
    double maxVal = 0;                                   // cuda::minMax(eig_, 0, &maxVal): NO mask
    { float m = eig[0]; for (float v : eig) m = std::max(m, v); maxVal = m; }
    const float thr = (float)(maxVal * quality);
    std::vector<int> cand;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            if (mask && !mask[(size_t)y * w + x]) continue;
            const float v = eig[(size_t)y * w + x];
            if (!(v > thr)) continue;
            float m = v;
            for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) m = std::fmax(eig[(size_t)(y + dy) * w + x + dx], m);
            if (v == m) cand.push_back(y * w + x);
        }
    if (cand.empty()) return;
    const bool tie_asc = dvo_get_variant("gftt_cuda_tie") == 1;
    std::sort(cand.begin(), cand.end(), [&](int p, int q) {      // EigGreater on the response; ties: declaration (b)
        return eig[p] > eig[q] ? true : eig[p] < eig[q] ? false : (tie_asc ? p < q : p > q); });
    const int total = (int)cand.size();
    if (min_dist < 1) {
        const int n = max_n > 0 ? std::min(max_n, total) : total;
        for (int i = 0; i < n; ++i) out.push_back({ (float)(cand[i] % w), (float)(cand[i] / w) });
        return;
    }
    const int cell = cv_round(min_dist);
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    std::vector<std::vector<P2>> grid((size_t)gw * gh);
    for (int i = 0; i < total; ++i) {
        const P2 p = { (float)(cand[i] % w), (float)(cand[i] / w) };
        const int xc = (int)(p.x / cell), yc = (int)(p.y / cell);
        const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
        bool good = true;
        for (int yy = y1; yy <= y2 && good; ++yy)
            for (int xx = x1; xx <= x2 && good; ++xx)
                for (const P2& m : grid[(size_t)yy * gw + xx]) {
                    const float dx = p.x - m.x, dy = p.y - m.y;
                    if (dx * dx + dy * dy < min_dist * min_dist) { good = false; break; }
                }
        if (good) {
            grid[(size_t)yc * gw + xc].push_back(p);
            out.push_back(p);
            if (max_n > 0 && (int)out.size() == max_n) break;
        }
    }
}

}  // namespace

extern "C" {
void dvo_min_eigen_cuda(const uint8_t* img, int w, int h, float* eig) { min_eigen_cuda(img, w, h, eig); }
void dvo_gftt_cuda(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n, double quality, double min_dist, float* out_xy, int* n_out) {
    std::vector<P2> out; gftt_cuda(img, mask, w, h, max_n, quality, min_dist, out);
    for (size_t i = 0; i < out.size(); ++i) { out_xy[2 * i] = out[i].x; out_xy[2 * i + 1] = out[i].y; }
    *n_out = (int)out.size();
}
}
