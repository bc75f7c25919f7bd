// gftt.hip — Shi-Tomasi corner selection (cv::goodFeaturesToTrack, blockSize 3, Sobel 3,
// min-distance grid) and the feature-bookkeeping kernels of FeatureTracker::TrackImage for gfx950.
//
// Reference call sites: front_end/background_tracker.cpp:79-96 (mask stamping with cv::circle,
// goodFeaturesToTrack, id assignment), front_end/instance_feature.cpp:352-392 (DetectNewFeature).
//
// Kernel 1  gftt_tile_kernel: one 256-thread workgroup per 64x16 tile.  u8 tile + 3 px halo in
//   LDS -> Sobel (fp32, kernel order of cv::Sobel with the 1/3060 scale folded into the
//   smoothing taps) -> covariance products -> 3x3 box sums in fp64 (fixed order) -> min
//   eigenvalue -> masked max (wave reduce + one atomicMax per workgroup) -> 3x3 local maxima
//   emitted as 16-byte candidate records.  The "mask" is never materialised: the filled discs
//   cv::circle would stamp around already-tracked points are evaluated analytically from a
//   per-tile culled point list, so the pass reads the image once and writes only candidates.
// Kernel 2  gftt_select_kernel: ONE workgroup.  Applies the quality threshold (needs the global
//   max, hence a second launch), orders candidates by (value desc, address desc) with an LDS
//   bitonic sort, and runs OpenCV's greedy min-distance acceptance on one wavefront: 64
//   candidates per step are tested against the accepted grid in parallel, conflicts inside
//   the step are resolved in lane order with ballots -> identical to the sequential loop.
//   Candidate sets larger than the LDS capacity are processed in value-ordered chunks.
//   Its epilogue appends the new corners to the tracker state (ids = global_id_count++).
#include "dv_internal.h"
#include "dev_once.h"
#include <cfloat>

#define TW 64
#define TH 16
#define IMG_W (TW + 6)
#define IMG_H (TH + 6)
#define IMG_PITCH 72
#define COV_W (TW + 4)
#define COV_H (TH + 4)
#define EIG_W (TW + 2)
#define EIG_H (TH + 2)
#define DISC_CAP 128

__device__ __forceinline__ int g_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}


// XCD-aware tile numbering: workgroups are dealt to the 8 XCDs round-robin, a 64-pixel tile row is HALF a 128-byte line and every tile reads a 3-pixel halo,
// so with the plain 2-D numbering neighbouring tiles (and both halves of every line) are fetched into different L2s.  Every XCD gets a contiguous run of
// tiles in raster order instead (the candidate order changes; the select kernel sorts by (value, address), so the result does not).
// lin = linear workgroup number of the launch, nt = tiles it covers -> the tile this workgroup takes
__device__ __forceinline__ int gftt_xcd_tile(int lin, int nt) {
    const int xc = lin & 7, k = lin >> 3;
    int start = 0;
    for (int y = 0; y < xc; ++y) start += nt > y ? (nt - y + 7) >> 3 : 0;
    return start + k;
}
// RULE 0: cv::goodFeaturesToTrack's response map and masked maximum.  RULE 1: the response map and the UNMASKED maximum of cv::cuda::GoodFeaturesToTrackDetector
// (cudaimgproc gftt.cpp / corners.cu; DetectShiTomasiCornersGpu, feature_utils.cpp:339-348 — TrackImageNaive's detector): the separable Sobel as a float
// multiply-add chain over the taps in order, the nine products of a block accumulated in raster order by fused multiply-adds, the closed form in float
// (choice D6: nvcc's default contraction, written out as fmaf here — the library is compiled with -ffp-contract=off).
template <int RULE>
__device__ __forceinline__ void gftt_tile_body(const GfttTileArgs& a, int tx, int ty) {
    __shared__ uint8_t s_img[IMG_H * IMG_PITCH];
    __shared__ float s_cov[3][COV_H][COV_W];
    __shared__ float s_eig[EIG_H][EIG_W];
    __shared__ int s_disc[DISC_CAP][2];
    __shared__ int s_ndisc;
    __shared__ unsigned s_max[4];
    __shared__ DvCand s_cand[TW * TH];          // every pixel of a plateau is a candidate (v >= all neighbours), so the cap is the tile
    __shared__ int s_ncand, s_base;
    const int tid = threadIdx.x;
    if (a.n_feat && a.max_cnt - *a.n_feat < a.min_new) return;     // no detection wanted this frame
    const int x0 = tx * TW, y0 = ty * TH;
    const int w = a.w, h = a.h;
    if (tid == 0) { s_ndisc = 0; s_ncand = 0; }
    for (int i = tid; i < IMG_H * IMG_PITCH; i += 256) {
        int r = i / IMG_PITCH, c = i - r * IMG_PITCH;
        if (c < IMG_W) {
            int y = g_reflect101(y0 - 3 + r, h), x = g_reflect101(x0 - 3 + c, w);
            s_img[i] = a.img[(size_t)y * a.pitch + x];
        }
    }
    __syncthreads();
    // ---- cull discs against this tile ----
    const int nd = (a.disc_pts && a.n_disc) ? *a.n_disc : 0;
    for (int i = tid; i < nd; i += 256) {
        float2 p = a.disc_pts[i];
        int cx = __float2int_rn(p.x), cy = __float2int_rn(p.y);
        if (cx + a.radius >= x0 && cx - a.radius < x0 + TW && cy + a.radius >= y0 && cy - a.radius < y0 + TH) {
            int k = atomicAdd(&s_ndisc, 1);
            if (k < DISC_CAP) { s_disc[k][0] = cx; s_disc[k][1] = cy; }
        }
    }
    // ---- Sobel + covariance products at logical positions [x0-2, x0+TW+2) x [y0-2, y0+TH+2) ----
    const float k1 = (float)(1.0 * (1.0 / (4.0 * 3.0 * 255.0))), k2 = (float)(2.0 * (1.0 / (4.0 * 3.0 * 255.0)));
    for (int i = tid; i < COV_H * COV_W; i += 256) {
        int r = i / COV_W, c = i - r * COV_W;
        int lx = x0 - 2 + c, ly = y0 - 2 + r;
        float xx = 0.f, xy = 0.f, yy = 0.f;
        if (lx >= -1 && lx <= w && ly >= -1 && ly <= h) {
            int rx = g_reflect101(lx, w), ry = g_reflect101(ly, h);      // boxFilter border: cov(-1) = cov(1)
            const uint8_t* p = &s_img[(ry - (y0 - 3)) * IMG_PITCH + (rx - (x0 - 3))];
            int a0 = p[-IMG_PITCH - 1], b0 = p[-IMG_PITCH], c0 = p[-IMG_PITCH + 1];
            int a1 = p[-1], c1 = p[1];
            int a2 = p[IMG_PITCH - 1], b2 = p[IMG_PITCH], c2 = p[IMG_PITCH + 1];
            float d0 = (float)(c0 - a0), d1 = (float)(c1 - a1), d2 = (float)(c2 - a2);
            if (RULE == 0) {
                float dx = (d0 + d2) * k1 + d1 * k2;
                float s0 = (k1 * (float)a0 + k2 * (float)b0) + k1 * (float)c0;
                float s2 = (k1 * (float)a2 + k2 * (float)b2) + k1 * (float)c2;
                float dy = s2 - s0;
                xx = dx * dx; xy = dx * dy; yy = dy * dy;
            } else {      // the derivatives themselves (xx <- Dx, xy <- Dy): the block sums below need them, not their products
                xx = fmaf(d2, k1, fmaf(d1, k2, d0 * k1));                                             // column chain over the exact row differences
                const float r0 = fmaf((float)c0, k1, fmaf((float)b0, k2, (float)a0 * k1));      // row chains [1 2 1] * scale of the rows above and below
                const float r2 = fmaf((float)c2, k1, fmaf((float)b2, k2, (float)a2 * k1));
                xy = r2 - r0;                                                                          // column kernel [-1 0 1]: -r0, + 0 * r1, + r2
            }
        }
        s_cov[0][r][c] = xx; s_cov[1][r][c] = xy; s_cov[2][r][c] = yy;
    }
    __syncthreads();
    // ---- min eigenvalue at [x0-1, x0+TW+1) x [y0-1, y0+TH+1) ----
    for (int i = tid; i < EIG_H * EIG_W; i += 256) {
        int r = i / EIG_W, c = i - r * EIG_W;
        int lx = x0 - 1 + c, ly = y0 - 1 + r;
        float e = -FLT_MAX;
        if (lx >= 0 && lx < w && ly >= 0 && ly < h) {
            if (RULE == 0) {
                float bx[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double rs0 = ((double)s_cov[k][r][c] + (double)s_cov[k][r][c + 1]) + (double)s_cov[k][r][c + 2];
                    double rs1 = ((double)s_cov[k][r + 1][c] + (double)s_cov[k][r + 1][c + 1]) + (double)s_cov[k][r + 1][c + 2];
                    double rs2 = ((double)s_cov[k][r + 2][c] + (double)s_cov[k][r + 2][c + 1]) + (double)s_cov[k][r + 2][c + 2];
                    bx[k] = (float)((rs0 + rs1) + rs2);
                }
                float A = bx[0] * 0.5f, B = bx[1], C = bx[2] * 0.5f;
                e = (A + C) - sqrtf((A - C) * (A - C) + B * B);
            } else {      // cornerMinEigenVal_kernel: a += dx * dx; b += dx * dy; c += dy * dy over the block in raster order
                float A = 0.f, B = 0.f, C = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float dx = s_cov[0][r + i][c + j], dy = s_cov[1][r + i][c + j];
                        A = fmaf(dx, dx, A); B = fmaf(dx, dy, B); C = fmaf(dy, dy, C);
                    }
                A *= 0.5f; C *= 0.5f;
                const float d = A - C;
                e = (A + C) - sqrtf(fmaf(d, d, B * B));
            }
        }
        s_eig[r][c] = e;
    }
    __syncthreads();
    // ---- own pixels: mask, max, local maxima ----
    const int ndisc = s_ndisc;
    unsigned vmax = 0;
    for (int i = tid; i < TH * TW; i += 256) {
        int r = i / TW, c = i - r * TW;
        int x = x0 + c, y = y0 + r;
        if (x >= w || y >= h) continue;
        float v = s_eig[r + 1][c + 1];
        if (a.eig_out) a.eig_out[(size_t)y * a.eig_pitch + x] = v;
        bool m = !a.in_mask || a.in_mask[(size_t)y * a.mask_pitch + x] != 0;
        if (m && nd > 0) {
            if (ndisc <= DISC_CAP) {
                for (int k = 0; k < ndisc; ++k) {
                    int dy = abs(y - s_disc[k][1]);
                    if (dy <= a.radius && abs(x - s_disc[k][0]) <= (int)a.hw[dy]) { m = false; break; }
                }
            } else {   // overflow of the culled list: test every disc
                for (int k = 0; k < nd; ++k) {
                    float2 p = a.disc_pts[k];
                    int dy = abs(y - __float2int_rn(p.y));
                    if (dy <= a.radius && abs(x - __float2int_rn(p.x)) <= (int)a.hw[dy]) { m = false; break; }
                }
            }
        }
        if (RULE == 1) vmax = max(vmax, f2ord(v));      // cuda::minMax(eig_, 0, &maxVal): no mask
        if (!m) continue;
        if (RULE == 0) vmax = max(vmax, f2ord(v));      // minMaxLoc(eig, 0, &maxVal, 0, 0, mask)
        if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1 || (RULE == 0 && v == 0.f)) continue;
        float nb[8] = { s_eig[r][c], s_eig[r][c + 1], s_eig[r][c + 2], s_eig[r + 1][c], s_eig[r + 1][c + 2],
                        s_eig[r + 2][c], s_eig[r + 2][c + 1], s_eig[r + 2][c + 2] };
        float mx = nb[0], mn = nb[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { mx = fmaxf(mx, nb[k]); mn = fminf(mn, nb[k]); }
        if (v >= mx) {          // candidates are collected in LDS: ONE global atomic per tile instead of one per candidate
            const int k = atomicAdd(&s_ncand, 1);
            if (k < TW * TH) { DvCand cd; cd.key = ((unsigned long long)f2ord(v) << 32) | (unsigned)(y * w + x); cd.min_nb = mn; cd.pad = 0; s_cand[k] = cd; }
        }
    }
    // workgroup max -> one atomic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = max(vmax, (unsigned)__shfl_xor((int)vmax, o));
    if ((tid & 63) == 0) s_max[tid >> 6] = vmax;
    __syncthreads();
    const int nc = min(s_ncand, TW * TH);
    if (tid == 0) {
        unsigned m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (m) atomicMax(a.max_ord, m);
        s_base = nc ? atomicAdd(a.n_cand, nc) : 0;
    }
    __syncthreads();
    for (int i = tid; i < nc; i += 256) { const int k = s_base + i; if (k < a.cand_cap) a.cand[k] = s_cand[i]; }
}
template <int RULE>
__global__ __launch_bounds__(256) void gftt_tile_kernel(GfttTileArgs a) {
    const int gx = gridDim.x, t = gftt_xcd_tile(blockIdx.y * gx + blockIdx.x, gridDim.x * gridDim.y);
    gftt_tile_body<RULE>(a, t % gx, t / gx);
}
// several images of ONE size in one launch (the front ends of a dv_batch group): the launch's tiles are numbered image-major, and every XCD takes a contiguous
// run of that sequence — whole images or raster runs of one
__global__ __launch_bounds__(256) void gftt_tile_multi_kernel(const GfttTileArgs* __restrict__ tab, int gx, int gy) {
    const int per = gx * gy, t = gftt_xcd_tile(blockIdx.x, gridDim.x), job = t / per, tt = t - job * per;
    const GfttTileArgs a = tab[job];
    const int tx = tt % gx, ty = tt / gx;
    if (tx * TW >= a.w || ty * TH >= a.h) return;      // (gx, gy) cover the largest image of the table: the objects' ROIs differ in size
    gftt_tile_body<0>(a, tx, ty);      // shared launches exist for DV_MODE_RAW only
}

// ---------------------------------------------------------------------------------------------
#define SEL_THREADS 1024
#define SEL_CAP 8192          // keys per LDS chunk
#define SEL_BINS 2048
#define SEL_MAX_ACC 1024
#define SEL_MAX_CELLS 32768


__device__ __forceinline__ void gftt_select_body(const GfttSelectArgs& a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);                    // SEL_CAP
    int* hist = reinterpret_cast<int*>(smem + SEL_CAP * 8);                                     // SEL_BINS + 1 (suffix sums)
    int* ctl = hist + SEL_BINS + 1;                                                             // 7 control words
    short* acc_x = reinterpret_cast<short*>(ctl + 7);                                           // SEL_MAX_ACC
    short* acc_y = acc_x + SEL_MAX_ACC;
    short* acc_next = acc_y + SEL_MAX_ACC;
    short* head = acc_next + SEL_MAX_ACC;                                                       // cells
    int& s_m = ctl[0]; int& s_acc = ctl[1]; int& s_lo = ctl[2]; int& s_hi = ctl[3]; int& s_done = ctl[4];
    const int tid = threadIdx.x;
    const int w = a.w, h = a.h;

    int max_n; bool want = true;
    if (a.n_feat) { max_n = a.max_cnt - *a.n_feat; want = max_n >= a.min_new; }
    else { max_n = a.max_n_host; if (max_n <= 0) max_n = SEL_MAX_ACC; }
    max_n = min(max_n, SEL_MAX_ACC);
    int n_cand = want ? min(*a.n_cand, a.cand_cap) : 0;
    if (want && *a.n_cand > a.cand_cap && tid == 0 && a.err_flag) atomicOr(a.err_flag, 1);
    const unsigned mo = *a.max_ord;
    const double maxVal = mo ? (double)ord2f(mo) : 0.0;
    const float thr = (float)(maxVal * a.quality);
    const float top = (float)maxVal;
    const float bscale = (top > thr) ? (float)SEL_BINS / (top - thr) : 0.f;
    const bool use_grid = a.min_dist >= 1.0;
    const int cell = use_grid ? (int)rint(a.min_dist) : 1;
    const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
    const double md2 = a.min_dist * a.min_dist;
    if (use_grid && gw * gh > SEL_MAX_CELLS) { if (tid == 0 && a.err_flag) atomicOr(a.err_flag, 2); n_cand = 0; }

    for (int i = tid; i <= SEL_BINS; i += SEL_THREADS) hist[i] = 0;
    if (use_grid) for (int i = tid; i < gw * gh; i += SEL_THREADS) head[i] = -1;
    if (tid == 0) { s_acc = 0; s_hi = SEL_BINS - 1; s_done = 0; }
    __syncthreads();
    // a candidate survives THRESH_TOZERO iff v > thr (v < 0 can never survive: thr >= 0 whenever maxVal >= 0)
    auto valid_bin = [&](const DvCand& c) -> int {
        float v = ord2f((unsigned)(c.key >> 32));
        if (!(v > thr)) return -1;
        if (a.rule == 0 && v < 0.f && !(c.min_nb > thr)) return -1;      // (rule 1, findCorners: eig > threshold and eig == max of the raw neighbourhood, nothing else)
        int b = (int)((v - thr) * bscale);
        return min(max(b, 0), SEL_BINS - 1);
    };
    // (eight candidates requested per trip: one load + wait per candidate made this loop ~12 dependent round trips for a 1280x720 frame)
    for (int i0 = tid; i0 < n_cand; i0 += 8 * SEL_THREADS) {
        DvCand cb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * SEL_THREADS; cb[u] = a.cand[i < n_cand ? i : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u * SEL_THREADS >= n_cand) continue;
            const int b = valid_bin(cb[u]);
            if (b >= 0) atomicAdd(&hist[b], 1);
        }
    }
    __syncthreads();
    // in-place inclusive suffix sums: hist[b] = #candidates with bin >= b ; hist[SEL_BINS] = 0
    for (int off = 1; off < SEL_BINS; off <<= 1) {
        int v0 = 0, v1 = 0;
        const int i0 = tid, i1 = tid + SEL_THREADS;
        if (i0 + off < SEL_BINS) v0 = hist[i0 + off];
        if (i1 + off < SEL_BINS) v1 = hist[i1 + off];
        __syncthreads();
        hist[i0] += v0; hist[i1] += v1;
        __syncthreads();
    }

    while (true) {
        // ---- next chunk of bins [lo, hi] from the top: smallest lo with count(lo..hi) <= SEL_CAP ----
        if (tid == 0) { if (s_hi < 0 || s_acc >= max_n || hist[0] - hist[s_hi + 1] == 0) s_done = 1; s_lo = s_hi; s_m = 0; }
        __syncthreads();
        if (s_done) break;
        const int hi = s_hi;
        {
            // Chunk size follows the demand: the greedy loop stops after max_n acceptances, which usually needs a few hundred of the strongest candidates, not
            // the 8192 a chunk can hold — and the bitonic sort below costs log^2 of the chunk.  Target 16 candidates per corner still wanted (at least 512);
            // the top bin is always taken whole (an error only if it alone exceeds the LDS capacity); further chunks follow if the target was too small.
            const int above = hist[hi + 1];
            const int target = min(SEL_CAP, max(512, 16 * (max_n - s_acc)));
            for (int b = tid; b <= hi; b += SEL_THREADS)
                if (hist[b] - above <= target) atomicMin(&s_lo, b);
            if (tid == 0 && hist[hi] - above > SEL_CAP && a.err_flag) atomicOr(a.err_flag, 4);   // one bin overflows a chunk
        }
        __syncthreads();
        const int lo = s_lo;
        for (int i0 = tid; i0 < n_cand; i0 += 8 * SEL_THREADS) {
            DvCand cb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * SEL_THREADS; cb[u] = a.cand[i < n_cand ? i : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u * SEL_THREADS >= n_cand) continue;
                const int b = valid_bin(cb[u]);
                if (b >= lo && b <= hi) { int k = atomicAdd(&s_m, 1); if (k < SEL_CAP) keys[k] = cb[u].key; }
            }
        }
        __syncthreads();
        const int m = min(s_m, SEL_CAP);
        int P = 1; while (P < m) P <<= 1;
        for (int i = m + tid; i < P; i += SEL_THREADS) keys[i] = 0ull;
        __syncthreads();
        // ---- bitonic sort, descending: (value desc, address desc) == cv greaterThanPtr ----
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += SEL_THREADS) {
                    int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    int l = i | j;
                    unsigned long long x = keys[i], y = keys[l];
                    bool desc = ((i & k) == 0);
                    if ((x < y) == desc) { keys[i] = y; keys[l] = x; }
                }
                __syncthreads();
            }
        }
        // ---- greedy min-distance acceptance on wave 0 (featureselect.cpp grid loop) ----
        if (tid < 64) {
            const int lane = tid;
            int acc = s_acc;
            for (int base = 0; base < m && acc < max_n; base += 64) {
                const int i = base + lane;
                bool good = i < m;
                int x = 0, y = 0;
                if (good) { unsigned idx = (unsigned)(keys[i] & 0xffffffffull); y = idx / w; x = idx - y * w; }
                if (good && use_grid) {
                    int xc = x / cell, yc = y / cell;
                    int x1 = max(0, xc - 1), y1 = max(0, yc - 1), x2 = min(gw - 1, xc + 1), y2 = min(gh - 1, yc + 1);
                    for (int yy = y1; yy <= y2 && good; ++yy)
                        for (int xx = x1; xx <= x2 && good; ++xx)
                            for (int q = head[yy * gw + xx]; q >= 0; q = acc_next[q]) {
                                float dx = (float)x - (float)acc_x[q], dy = (float)y - (float)acc_y[q];
                                if ((double)(dx * dx + dy * dy) < md2) { good = false; break; }
                            }
                }
                unsigned long long pending = __ballot(good);
                while (pending && acc < max_n) {
                    const int l = __ffsll((long long)pending) - 1;
                    const int ax = __shfl(x, l), ay = __shfl(y, l);
                    if (lane == l) {
                        acc_x[acc] = (short)ax; acc_y[acc] = (short)ay;
                        if (use_grid) { int cidx = (ay / cell) * gw + ax / cell; acc_next[acc] = head[cidx]; head[cidx] = (short)acc; }
                        good = false;
                    } else if (good && use_grid) {
                        float dx = (float)x - (float)ax, dy = (float)y - (float)ay;
                        if ((double)(dx * dx + dy * dy) < md2) good = false;
                    }
                    ++acc;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    pending = __ballot(good);
                }
            }
            if (lane == 0) { s_acc = acc; s_hi = lo - 1; }
        }
        __syncthreads();
    }
    __syncthreads();
    const int acc = s_acc;
    if (a.out_xy) {
        for (int i = tid; i < acc; i += SEL_THREADS) a.out_xy[i] = make_float2((float)acc_x[i], (float)acc_y[i]);
        if (tid == 0) *a.n_out = acc;
    }
    if (a.has_tr) {   // background_tracker.cpp:92-96: append, id = global_id_count++, track_cnt = 1
        const int n0 = *a.tr.n_feat;
        const bool own_ids = a.has_tr == 1;
        const unsigned id0 = own_ids ? *a.tr.next_id : 0u;
        __syncthreads();
        for (int i = tid; i < acc; i += SEL_THREADS) {
            a.tr.curr_pts[n0 + i] = make_float2((float)acc_x[i], (float)acc_y[i]);
            if (own_ids) a.tr.ids[n0 + i] = id0 + i;
            a.tr.track_cnt[n0 + i] = 1;
            a.tr.tracked[n0 + i] = 0;
            a.tr.prev_rvalid[n0 + i] = 0;
        }
        if (tid == 0) {
            *a.tr.n_feat = n0 + acc;
            if (own_ids) *a.tr.next_id = id0 + acc; else { a.id_slot[0] = n0; a.id_slot[1] = acc; }
        }
    }
}
// ids of the corners the jobs of ONE select launch appended (has_tr == 2), drawn from their common counter in job order: exactly what the jobs' own epilogues do
// when they run one after the other (InstFeat::global_id_count, one static counter for background and object features)
__global__ __launch_bounds__(256) void gftt_assign_ids_kernel(const GfttSelectArgs* __restrict__ tab, int n_jobs) {
    unsigned* next = tab[0].tr.next_id;
    unsigned base = *next;
    for (int j = 0; j < n_jobs; ++j) {
        const int n0 = tab[j].id_slot[0], acc = tab[j].id_slot[1];
        uint32_t* ids = tab[j].tr.ids;
        for (int i = threadIdx.x; i < acc; i += 256) ids[n0 + i] = base + i;
        base += (unsigned)acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) *next = base;
}
__global__ __launch_bounds__(SEL_THREADS) void gftt_select_kernel(GfttSelectArgs a) { gftt_select_body(a); }
// one workgroup per job (blockIdx.x): the corner selections of a dv_batch group's front ends in one launch
__global__ __launch_bounds__(SEL_THREADS) void gftt_select_multi_kernel(const GfttSelectArgs* __restrict__ tab) {
    const GfttSelectArgs a = tab[blockIdx.x];
    gftt_select_body(a);
}

static size_t select_smem_bytes() {
    return (size_t)SEL_CAP * 8 + (SEL_BINS + 1 + 7) * 4 + 3 * SEL_MAX_ACC * 2 + SEL_MAX_CELLS * 2;
}

void dv_launch_gftt_tile(const GfttTileArgs& a, hipStream_t s) {
    dim3 grid((a.w + TW - 1) / TW, (a.h + TH - 1) / TH);
    if (a.rule == DV_GFTT_RULE_CUDA) hipLaunchKernelGGL(gftt_tile_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gftt_tile_kernel<0>, grid, dim3(256), 0, s, a);
}

void dv_launch_gftt_tile_multi(const GfttTileArgs* tab_dev, int n_jobs, int w, int h, hipStream_t s) {
    if (n_jobs <= 0) return;
    const int gx = (w + TW - 1) / TW, gy = (h + TH - 1) / TH;
    hipLaunchKernelGGL(gftt_tile_multi_kernel, dim3(gx * gy * n_jobs), dim3(256), 0, s, tab_dev, gx, gy);
}
void dv_launch_gftt_assign_ids(const GfttSelectArgs* tab_dev, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(gftt_assign_ids_kernel, dim3(1), dim3(256), 0, s, tab_dev, n_jobs);
}
int dv_launch_gftt_select_multi(const GfttSelectArgs* tab_dev, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return 0;
    static DevOnce once;
    if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gftt_select_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)select_smem_bytes()) != hipSuccess; })) return -1;
    hipLaunchKernelGGL(gftt_select_multi_kernel, dim3(n_jobs), dim3(SEL_THREADS), select_smem_bytes(), s, tab_dev);
    return 0;
}

int dv_launch_gftt_select(const GfttSelectArgs& a, hipStream_t s) {
    static DevOnce once;
    if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(gftt_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)select_smem_bytes()) != hipSuccess; })) return -1;
    hipLaunchKernelGGL(gftt_select_kernel, dim3(1), dim3(SEL_THREADS), select_smem_bytes(), s, a);
    return 0;
}
