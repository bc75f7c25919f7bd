// lk_cuda.cpp — CPU ORACLE (test infrastructure, NOT the product) of the reference's GPU tracker:
//   FeatureTrackByLKGpu                          front_end/feature_utils.cpp:83-163      forward + backward cv::cuda::SparsePyrLKOpticalFlow, |p - p_rev| <= 1.0, InBorder
//   cv::cuda::SparsePyrLKOpticalFlow::create(Size(21, 21), 3, 30[, useInitialFlow])       front_end/background_tracker.cpp:34-36 (forward, and backward with the
//                                                                                         previous points as the initial flow)
// used where the reference uses it: the temporal and the right-image tracking of TrackImageNaive (TrackLeftGPU / TrackRightGPU, instance_feature.cpp:191-310) and the
// right image of TrackSemanticImage (background_tracker.cpp:797-798).
//
// OpenCV 3.4.16's cudaoptflow / cudawarping modules are NOT under /root/reference and the reference holds no vectors for them: restated from the published sources
// (modules/cudaoptflow/src/pyrlk.cpp, src/cuda/pyrlk.cu; modules/cudawarping/src/cuda/pyr_down.cu) — PARITY UNPINNED, more so than the CPU pieces, because two things
// the library leaves to the platform are not recoverable and are FIXED here by declaration:
//   (a) texture filtering.  The kernels sample 8-bit images through 2-D textures with cudaFilterModeLinear, cudaAddressModeClamp, cudaReadModeNormalizedFloat and
//       unnormalised coordinates.  The CUDA programming guide defines tex(x, y) = (1-a)(1-b) T[i, j] + a (1-b) T[i+1, j] + (1-a) b T[i, j+1] + a b T[i+1, j+1] with
//       xB = x - 0.5, i = floor(xB), a = frac(xB) "stored in 9-bit fixed point format with 8 bits of fractional value".  Here: T = u8 / 255.0f, a and b rounded to the
//       nearest 1/256 (ties up), the four products summed left to right in float.
//   (b) floating-point contraction.  nvcc fuses a * b + c where it sees fit; which sites of pyrlk.cu it fused in the reference's build is unknown.  Here: no
//       contraction anywhere (the product is compiled with -ffp-contract=off).
// Everything else follows the kernel source: one 16 x 16 thread block per point with a 2 x 2 pixel patch per thread (calcPatchSize for a 21 x 21 window), Scharr
// derivatives on the fly, per-thread partial sums in patch order, the block reduction of cudev's GenericOptimized32<256> (32-lane shuffle trees, then one over the
// eight warp results), D < FLT_EPSILON as the only conditioning test (NO minimum-eigenvalue test), <= iters iterations with |dx|, |dy| < 0.01 as the stop rule, the
// half-window 10 = (21 - 1) / 2 as an integer, status cleared at level 0 only — and a level that bails out early leaves nextPts untouched, so the next finer level
// doubles a coarse-level coordinate once too little (kept: it is what the library does).
// cuda::pyrDown on 8-bit images computes the same 5 x 5 binomial sum as cv::pyrDown (exactly: every partial sum is a multiple of 1/256 below 256) but rounds it with
// saturate_cast<uchar>(float) = round-half-to-EVEN, where the CPU code does (sum + 128) >> 8 (half up): the two pyramids differ at the pixels whose sum ends in .5.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include "dvo.h"

namespace {

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

struct ImgC { int w = 0, h = 0; std::vector<uint8_t> d; };

void pyr_down_cuda(const uint8_t* src, int w, int h, ImgC& dst) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    dst.w = dw; dst.h = dh; dst.d.resize((size_t)dw * dh);
    static const int k[5] = { 1, 4, 6, 4, 1 };
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int sum = 0;
            for (int j = -2; j <= 2; ++j) {
                const int sy = reflect101(2 * y + j, h);
                int row = 0;
                for (int i = -2; i <= 2; ++i) row += k[i + 2] * src[(size_t)sy * w + reflect101(2 * x + i, w)];
                sum += k[j + 2] * row;
            }
            int q = sum >> 8; const int r = sum & 255;             // saturate_cast<uchar>(float sum / 256): __float2int_rn
            if (r > 128 || (r == 128 && (q & 1))) ++q;
            dst.d[(size_t)y * dw + x] = (uint8_t)std::min(q, 255);
        }
}

// tex2D(linear, clamp, normalised-float read) at unnormalised (x, y)
inline float tex_read(const ImgC& I, float x, float y) {
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = std::floor(xb), fy = std::floor(yb);
    const int i = (int)fx, j = (int)fy;
    const float a = std::floor((xb - fx) * 256.f + 0.5f) * (1.f / 256.f), b = std::floor((yb - fy) * 256.f + 0.5f) * (1.f / 256.f);
    auto T = [&](int xi, int yj) { xi = std::min(std::max(xi, 0), I.w - 1); yj = std::min(std::max(yj, 0), I.h - 1); return (float)I.d[(size_t)yj * I.w + xi] / 255.0f; };
    float v = (1.f - a) * (1.f - b) * T(i, j);
    v = v + a * (1.f - b) * T(i + 1, j);
    v = v + (1.f - a) * b * T(i, j + 1);
    v = v + a * b * T(i + 1, j + 1);
    return v;
}

// cudev::blockReduce<256> (GenericOptimized32): shuffle-down trees inside the eight 32-lane warps, then over the eight warp results
inline float block_reduce_256(float* v) {
    float w[8];
    for (int wp = 0; wp < 8; ++wp) {
        float* l = v + 32 * wp;
        for (int d = 16; d >= 1; d >>= 1) for (int t = 0; t < d; ++t) l[t] = l[t] + l[t + d];      // lane t += lane t + d (the lanes that matter read in-range values)
        w[wp] = l[0];
    }
    for (int d = 4; d >= 1; d >>= 1) for (int t = 0; t < d; ++t) w[t] = w[t] + w[t + d];
    return w[0];
}

const int WIN = 21, HALF = (WIN - 1) / 2, BX = 16, BY = 16;

// one pyramid level of one point: pyrlk::sparseKernel<1, 2, 2, false, uchar>.  Returns false when the kernel `return`s before writing nextPts.
bool sparse_level(const ImgC& I, const ImgC& J, float px, float py, float& nx, float& ny, uint8_t& status, int level, int iters) {
    const int rows = I.h, cols = I.w;
    float prx = px * (1.0f / (float)(1 << level)), pry = py * (1.0f / (float)(1 << level));
    if (prx < 0 || prx >= cols || pry < 0 || pry >= rows) { if (level == 0) status = 0; return false; }
    prx -= (float)HALF; pry -= (float)HALF;
    static thread_local float Ip[256][2][2], Dx[256][2][2], Dy[256][2][2], s1[256], s2[256], s3[256];
    for (int ty = 0; ty < BY; ++ty) for (int tx = 0; tx < BX; ++tx) {
        const int tid = ty * BX + tx;
        float a11 = 0, a12 = 0, a22 = 0;
        for (int yb = ty, i = 0; yb < WIN; yb += BY, ++i)
            for (int xb = tx, j = 0; xb < WIN; xb += BX, ++j) {
                const float x = prx + xb + 0.5f, y = pry + yb + 0.5f;
                Ip[tid][i][j] = tex_read(I, x, y);
                const float dIdx = 3.0f * tex_read(I, x + 1, y - 1) + 10.0f * tex_read(I, x + 1, y) + 3.0f * tex_read(I, x + 1, y + 1) -
                                   (3.0f * tex_read(I, x - 1, y - 1) + 10.0f * tex_read(I, x - 1, y) + 3.0f * tex_read(I, x - 1, y + 1));
                const float dIdy = 3.0f * tex_read(I, x - 1, y + 1) + 10.0f * tex_read(I, x, y + 1) + 3.0f * tex_read(I, x + 1, y + 1) -
                                   (3.0f * tex_read(I, x - 1, y - 1) + 10.0f * tex_read(I, x, y - 1) + 3.0f * tex_read(I, x + 1, y - 1));
                Dx[tid][i][j] = dIdx; Dy[tid][i][j] = dIdy;
                a11 += dIdx * dIdx; a12 += dIdx * dIdy; a22 += dIdy * dIdy;
            }
        s1[tid] = a11; s2[tid] = a12; s3[tid] = a22;
    }
    float A11 = block_reduce_256(s1), A12 = block_reduce_256(s2), A22 = block_reduce_256(s3);
    float D = A11 * A22 - A12 * A12;
    if (D < FLT_EPSILON) { if (level == 0) status = 0; return false; }
    D = 1.f / D;
    A11 *= D; A12 *= D; A22 *= D;
    float qx = nx * 2.f, qy = ny * 2.f;
    qx -= (float)HALF; qy -= (float)HALF;
    for (int k = 0; k < iters; ++k) {
        if (qx < -(float)HALF || qx >= cols || qy < -(float)HALF || qy >= rows) { if (level == 0) status = 0; return false; }
        for (int ty = 0; ty < BY; ++ty) for (int tx = 0; tx < BX; ++tx) {
            const int tid = ty * BX + tx;
            float b1 = 0, b2 = 0;
            for (int y = ty, i = 0; y < WIN; y += BY, ++i)
                for (int x = tx, j = 0; x < WIN; x += BX, ++j) {
                    const float Jv = tex_read(J, qx + x + 0.5f, qy + y + 0.5f);
                    const float diff = (Jv - Ip[tid][i][j]) * 32.0f;
                    b1 += diff * Dx[tid][i][j];
                    b2 += diff * Dy[tid][i][j];
                }
            s1[tid] = b1; s2[tid] = b2;
        }
        const float b1 = block_reduce_256(s1), b2 = block_reduce_256(s2);
        const float dx = A12 * b2 - A22 * b1, dy = A12 * b1 - A11 * b2;
        qx += dx; qy += dy;
        if (std::fabs(dx) < 0.01f && std::fabs(dy) < 0.01f) break;
    }
    nx = qx + (float)HALF; ny = qy + (float)HALF;
    return true;
}

// PyrLKOpticalFlowBase::sparse (pyrlk.cpp): nextPts = (useInitialFlow ? nextPts : prevPts) * (1 / 2^maxLevel / 2); status = 1; levels maxLevel .. 0
void sparse_lk(const std::vector<ImgC>& A, const std::vector<ImgC>& B, const float* pa, int n, int max_level, int iters, bool use_initial, float* pb, uint8_t* st) {
    const float scale = (float)(1.0 / (1 << max_level) / 2.0);
    for (int p = 0; p < n; ++p) {
        float nx = (use_initial ? pb[2 * p] : pa[2 * p]) * scale, ny = (use_initial ? pb[2 * p + 1] : pa[2 * p + 1]) * scale;
        uint8_t s = 1;
        for (int level = max_level; level >= 0; --level) sparse_level(A[level], B[level], pa[2 * p], pa[2 * p + 1], nx, ny, s, level, iters);
        pb[2 * p] = nx; pb[2 * p + 1] = ny; st[p] = s;
    }
}

void build_pyr(const uint8_t* img, int w, int h, int max_level, std::vector<ImgC>& P) {
    P.resize(max_level + 1);
    P[0].w = w; P[0].h = h; P[0].d.assign(img, img + (size_t)w * h);
    for (int l = 1; l <= max_level; ++l) pyr_down_cuda(P[l - 1].d.data(), P[l - 1].w, P[l - 1].h, P[l]);
}

inline bool in_border(float x, float y, int rows, int cols) {          // InBorder (feature_utils.h:68-74): cvRound
    const int ix = (int)std::nearbyint(x), iy = (int)std::nearbyint(y);
    return 1 <= ix && ix < cols - 1 && 1 <= iy && iy < rows - 1;
}

}  // namespace

extern "C" {

void dvo_pyr_down_cuda(const uint8_t* src, int w, int h, uint8_t* dst) {
    ImgC d; pyr_down_cuda(src, w, h, d);
    std::memcpy(dst, d.d.data(), d.d.size());
}
float dvo_tex_read(const uint8_t* img, int w, int h, float x, float y) {
    ImgC I; I.w = w; I.h = h; I.d.assign(img, img + (size_t)w * h);
    return tex_read(I, x, y);
}
// cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), max_level, iters, use_initial)->calc(img_a, img_b, pts_a, pts_b, status)
void dvo_lk_cuda(const uint8_t* img_a, const uint8_t* img_b, int w, int h, const float* pts_a, int n, int max_level, int iters, int use_initial, float* pts_b, uint8_t* status) {
    std::vector<ImgC> A, B;
    build_pyr(img_a, w, h, max_level, A); build_pyr(img_b, w, h, max_level, B);
    sparse_lk(A, B, pts_a, n, max_level, iters, use_initial != 0, pts_b, status);
}
// FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163)
void dvo_track_by_lk_gpu(const uint8_t* img1, const uint8_t* img2, int w, int h, const float* pts1, int n, int flow_back, float* pts2, uint8_t* status) {
    std::vector<ImgC> A, B;
    build_pyr(img1, w, h, 3, A); build_pyr(img2, w, h, 3, B);
    sparse_lk(A, B, pts1, n, 3, 30, false, pts2, status);
    if (flow_back) {
        std::vector<float> rev(pts1, pts1 + 2 * (size_t)n);          // d_reverse_pts = d_prevPts: the backward tracker starts from the previous points (useInitialFlow)
        std::vector<uint8_t> rst((size_t)n);
        sparse_lk(B, A, pts2, n, 3, 30, true, rev.data(), rst.data());
        for (int i = 0; i < n; ++i) {
            const float dx = pts1[2 * i] - rev[2 * i], dy = pts1[2 * i + 1] - rev[2 * i + 1];
            status[i] = (status[i] && rst[i] && std::sqrt(dx * dx + dy * dy) <= 1.f) ? 1 : 0;      // PointDistance(...) <= 1.
        }
    }
    for (int i = 0; i < n; ++i) if (status[i] && !in_border(pts2[2 * i], pts2[2 * i + 1], h, w)) status[i] = 0;
}

}  // extern "C"
