// lk.hip — pyramidal Lucas-Kanade sparse optical flow for gfx950.
//
// Semantics: cv::calcOpticalFlowPyrLK (OpenCV 3.4.16, winSize 21x21, 14-bit fixed-point
// bilinear weights, Scharr derivatives, minEigThreshold 1e-4) as called by the reference's
// FeatureTrackByLK (dynamic_vins/src/front_end/feature_utils.cpp:35-69).
//
// MI355X mapping: ONE 64-lane wavefront per feature point, no workgroup barriers.
//   * lane -> (window row r = lane/3, 7-pixel segment lane%3): 63 lanes cover the 21x21 window;
//   * per level the 24x24 I tile (window + 1 px Scharr halo + 1 px bilinear) and a 32x32 J tile
//     are staged once in LDS with 4-byte aligned dword loads (reflect-101 byte path only for
//     tiles that touch the image border); the J tile is re-centred only when the iterate
//     drifts more than 5 px;
//   * the interpolated patch I, Ix, Iy stays in registers (21 ints/lane) for all iterations;
//   * A11,A12,A22,b1,b2 are exact integer sums: int32 butterflies over 8 lanes, widened to
//     int64 for the last three steps -> results are independent of reduction order and
//     bit-identical to the CPU oracle;
//   * forward (4 levels) + backward (2 levels, initial flow) + distance test + InBorder test
//     run in the same wave, so one launch replaces two calcOpticalFlowPyrLK calls and the
//     host loop at feature_utils.cpp:55-66.
// Float math is compiled with -ffp-contract=off and uses correctly rounded div/sqrt.
#include "dv_internal.h"
#include "wave_dpp.h"
#ifdef LK_TS
// debug build only: iteration counts and wall-clock (100 MHz) of the LK waves.  [0] iterations, [1] level passes, [3] max duration of one point,
// [4] sum of loop time, [5] sum of pre-loop (staging, gradients, A) time, [6] points, [7] sum of point durations, [8] level passes whose I tile took the
// border (byte-wise, reflect-101) path, [9] their pre-loop time, [10] J tiles staged, [11] J tiles on the border path, [12] max iterations of a level pass, [13] sum of J staging time
__device__ unsigned long long lk_dbg[16];
extern "C" int dv_debug_lk_ts(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lk_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(lk_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define LKADD(k, v) do { if (lane == 0) atomicAdd(&lk_dbg[k], (unsigned long long)(v)); } while (0)
#define LKMAX(k, v) do { if (lane == 0) atomicMax(&lk_dbg[k], (unsigned long long)(v)); } while (0)
#define LKNOW() wall_clock64()
#else
#define LKADD(k, v) do {} while (0)
#define LKMAX(k, v) do {} while (0)
#define LKNOW() 0ll
#endif
#include <cfloat>

#define WIN DV_LK_WIN
#define IT_ROWS 24
#define IT_PITCH 28
#define JT_ROWS 32
#define JT_PITCH 36
#define JT_MARGIN 5

// image bytes are read through explicit global-address-space pointers: the level descriptors may come out of LDS, and a generic (flat) load counts on the LDS
// counter too — every later LDS wait would then wait for the prefetched tiles
typedef const uint8_t __attribute__((address_space(1)))* lk_gp8;
typedef const uint32_t __attribute__((address_space(1)))* lk_gp32;
__device__ __forceinline__ lk_gp8 lk_g8(const uint8_t* p) { return (lk_gp8)(unsigned long long)p; }
__device__ __forceinline__ lk_gp32 lk_g32(const uint8_t* p) { return (lk_gp32)(unsigned long long)p; }
__device__ __forceinline__ int lk_reflect101(int p, int len) {
    if (len == 1) return 0;
    if (p < 0) p = -p;                                   // one reflection is the rule (a tile overhangs the image by < 27 px); the loop only runs for images smaller than that
    if (p >= len) p = 2 * len - 2 - p;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// Stage rows [y0, y0+rows) x bytes [ax0, ax0+pitch) of level L into LDS (ax0 multiple of 4).
// need_x0/need_cols describe the columns that are actually consumed; if they and the rows lie
// inside the image the aligned dword path is taken (over-read bytes are never used).
__device__ __forceinline__ void lk_stage_tile(uint8_t* lds, int pitch, int rows, const DvLevel& L, int ax0, int y0,
                                              int need_x0, int need_cols, int lane) {
    const bool fast = need_x0 >= -L.apron && need_x0 + need_cols <= L.w + L.apron && y0 >= -L.apron && y0 + rows <= L.h + L.apron;      // inside the image or its reflect-101 apron
    if (fast) {
        const int ndw = pitch >> 2;
        uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
        // eight dwords requested per trip (the whole I tile is 3 per lane, the J tile 5): a load + wait per dword made a tile 3 resp. 5 dependent round trips
        for (int i0 = lane; i0 < rows * ndw; i0 += 8 * 64) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u < rows * ndw ? i0 + 64 * u : i0;
                const int r = i / ndw, c = i - r * ndw;
                v[u] = *lk_g32(L.p + (ptrdiff_t)(y0 + r) * L.pitch + ax0 + 4 * c);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + 64 * u < rows * ndw) l32[i0 + 64 * u] = v[u];
        }
    } else {
        // border path (17 - 22 % of the tiles; most tiles of the coarse levels): every reflected byte of the tile requested in one trip, then stored — one load +
        // wait per byte made a J tile 18 dependent round trips, eight per trip still three (the slowest points of a launch are the border ones: 74 -> 65 us per launch)
        for (int i0 = lane; i0 < rows * pitch; i0 += 18 * 64) {      // a whole J tile (18 bytes per lane) per trip
            uint8_t v[18];
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int i = i0 + 64 * u < rows * pitch ? i0 + 64 * u : i0;
                const int r = i / pitch, c = i - r * pitch;
                v[u] = *lk_g8(L.p + (size_t)lk_reflect101(y0 + r, L.h) * L.pitch + lk_reflect101(ax0 + c, L.w));
            }
#pragma unroll
            for (int u = 0; u < 18; ++u) if (i0 + 64 * u < rows * pitch) lds[i0 + 64 * u] = v[u];
        }
    }
}

// Tiles of the NEXT level pass, requested while the current pass iterates (the staging round trips — one per tile, ~1.5 us each on an otherwise idle CU —
// were a third of a point's dependent chain).  The I tile of a pass is known in advance (it depends on the point and the level only); the J tile is
// speculative: centred on where the current iterate would put it, checked against the real start of the next pass (its +-5 px margin absorbs what the
// remaining iterations of this level move, x2).  Only tiles that lie inside the image are prefetched (the aligned dword path); the registers are stored
// to LDS when the pass begins.
#define LK_PRE_I 3      // dwords per lane: 24 rows x 7 dwords = 168 <= 3 x 64
#define LK_PRE_J 5      // 32 rows x 9 dwords = 288 <= 5 x 64
struct LkPre { int i_ok, j_ok, jx0, jy0; };
__device__ __forceinline__ bool lk_tile_inside(const DvLevel& L, int x0, int cols, int y0, int rows) { return x0 >= -L.apron && x0 + cols <= L.w + L.apron && y0 >= -L.apron && y0 + rows <= L.h + L.apron; }
// requests tile rows [y0, y0 + rows) x bytes [ax0, ax0 + pitch) of level L as dwords if the consumed columns [x0, x0 + cols) and the rows lie inside the image
// (returns 1; else 0: border tiles are staged when they are needed — prefetching their 11 + 18 reflected bytes per lane as well cost 48 VGPRs, one wave per
// SIMD less, and 4 % of the multi-sequence throughput for 10 % of this kernel's latency)
template <int N>
__device__ __forceinline__ int lk_prefetch(uint32_t (&v)[N], int pitch, int rows, const DvLevel& L, int ax0, int y0, int x0, int cols, int lane) {
    if (!lk_tile_inside(L, x0, cols, y0, rows)) return 0;
    const int ndw = pitch >> 2, total = rows * ndw;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int i = lane + 64 * u < total ? lane + 64 * u : lane;
        const int r = i / ndw, c = i - r * ndw;
        v[u] = *lk_g32(L.p + (ptrdiff_t)(y0 + r) * L.pitch + ax0 + 4 * c);
    }
    return 1;
}
template <int N>
__device__ __forceinline__ void lk_commit(uint8_t* lds, const uint32_t (&v)[N], int pitch, int rows, int lane) {
    const int total = rows * (pitch >> 2);
    uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
    for (int u = 0; u < N; ++u) if (lane + 64 * u < total) l32[lane + 64 * u] = v[u];
}
// what the next pass will be: its images, level, previous point (known or not) and how its start iterate follows from this pass
struct LkNext { DvLevel I, J; int valid, level; float2 prev; int prev_known; int mode; };      // mode 0: next finer level of the same direction; 1: first level of the backward pass (initial flow = prev)

__device__ __forceinline__ long long lk_wave_sum(int v) {
    // |v| * 8 < 2^31 for every quantity summed here (see DESIGN.md "LK exact sums")
    // on the DPP path: the ds_bpermute butterfly (nine dependent LDS-crossbar round trips per sum, two sums per iteration) was the longest link of the
    // iteration's dependent chain.  Integer sums: the result is the same whatever the tree.
    return wave_sum_i32_wide(v);
}

__device__ __forceinline__ int lk_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
// Integer products below go through __mul24 (v_mul_i32_i24, full rate; v_mul_lo_u32 is quarter rate): every operand is far inside 24 bits — pixels < 2^8,
// bilinear weights <= 2^14, Scharr sums < 2^13, interpolated values and differences < 2^15 — and the low 32 bits of the product are exact.

__device__ __forceinline__ void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    w10 = __float2int_rn((1.f - a) * b * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

// One pyramid level of LKTrackerInvoker for the wave's point.  All control flow is wave-uniform.
// next is in/out (nextPts[ptidx]); status is cleared only at level 0.
__device__ __forceinline__ void lk_level(const DvLevel& I, const DvLevel& J, int level, int max_level, float2 prev, float2& next,
                         bool& status, int max_count, double eps_sq, bool use_initial, uint8_t* sI, uint8_t* sJ, int lane, LkPre& pre, uint32_t (&piv)[LK_PRE_I], uint32_t (&pjv)[LK_PRE_J], const LkNext& nx) {
    // tiles prefetched for THIS pass (by the previous one); whatever path leaves this function, `pre` then describes the NEXT pass
    const int have_i = pre.i_ok, have_j = pre.j_ok, pjx0 = pre.jx0, pjy0 = pre.jy0;
    pre.i_ok = pre.j_ok = 0;
    const long long lk_t0 = LKNOW(); (void)lk_t0;
    const float half = (WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float lscale = (float)(1. / (1 << level));
    float2 prevPt = make_float2(prev.x * lscale, prev.y * lscale);
    float2 nextPt;
    if (level == max_level) nextPt = use_initial ? make_float2(next.x * lscale, next.y * lscale) : prevPt;
    else nextPt = make_float2(next.x * 2.f, next.y * 2.f);
    next = nextPt;

    prevPt.x -= half; prevPt.y -= half;
    const int ipx = (int)floorf(prevPt.x), ipy = (int)floorf(prevPt.y);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) { if (level == 0) status = false; return; }
    int w00, w01, w10, w11;
    lk_weights(prevPt.x - ipx, prevPt.y - ipy, w00, w01, w10, w11);

    // ---- stage the I tile: rows ipy-1..ipy+22, cols ipx-1..ipx+22 ----
    const int itx0 = ipx - 1, iax0 = itx0 & ~3, ioff = itx0 - iax0;
    if (have_i) lk_commit(sI, piv, IT_PITCH, IT_ROWS, lane);
    else lk_stage_tile(sI, IT_PITCH, IT_ROWS, I, iax0, ipy - 1, itx0, 24, lane);
    const bool lk_slow_i = !(itx0 >= 0 && itx0 + 24 <= I.w && ipy - 1 >= 0 && ipy - 1 + IT_ROWS <= I.h); (void)lk_slow_i;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const bool active = lane < 63;
    const int r = active ? lane / 3 : 0, c0 = active ? (lane - r * 3) * 7 : 0;
    int Iv[7], Ixv[7], Iyv[7];
    {
        // tile coordinates of window pixel (x=c0, y=r): row r+1, col c0+1+ioff
        const uint8_t* T = sI + (r + 1) * IT_PITCH + (c0 + 1 + ioff);
        // column-wise Scharr partials for rows y=r (j=0) and y=r+1 (j=1), cols c0-1..c0+8
        int t0[2][10], t1[2][10];
        // ten bytes of each of the four rows as one 8-byte and one 2-byte LDS read (gfx950 reads LDS unaligned): 8 LDS instructions instead of 40 byte reads
        uint64_t q8[4]; uint16_t q2[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { __builtin_memcpy(&q8[rr], T + (rr - 1) * IT_PITCH - 1, 8); __builtin_memcpy(&q2[rr], T + (rr - 1) * IT_PITCH + 7, 2); }
        auto byte_of = [&](int rr, int k) -> int { return k < 8 ? (int)((q8[rr] >> (8 * k)) & 0xff) : (int)((q2[rr] >> (8 * (k - 8))) & 0xff); };
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            int a = byte_of(0, k), b = byte_of(1, k), c = byte_of(2, k), d = byte_of(3, k);
            t0[0][k] = (a + c) * 3 + b * 10;  t1[0][k] = c - a;
            t0[1][k] = (b + d) * 3 + c * 10;  t1[1][k] = d - b;
        }
        int gx[2][8], gy[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ay = ipy + r + j;
            const bool yin = ay >= 0 && ay < I.h;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ax = ipx + c0 + i;
                const bool in = yin && ax >= 0 && ax < I.w;          // derivative image has a zero border
                gx[j][i] = in ? t0[j][i + 2] - t0[j][i] : 0;
                gy[j][i] = in ? (t1[j][i + 2] + t1[j][i]) * 3 + t1[j][i + 1] * 10 : 0;
            }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            Iv[i] = lk_descale(__mul24(byte_of(1, i + 1), w00) + __mul24(byte_of(1, i + 2), w01) + __mul24(byte_of(2, i + 1), w10) + __mul24(byte_of(2, i + 2), w11), 14 - 5);      // T[i] = row r+1, col k = i+1 of the bytes read above
            Ixv[i] = lk_descale(__mul24(gx[0][i], w00) + __mul24(gx[0][i + 1], w01) + __mul24(gx[1][i], w10) + __mul24(gx[1][i + 1], w11), 14);
            Iyv[i] = lk_descale(__mul24(gy[0][i], w00) + __mul24(gy[0][i + 1], w01) + __mul24(gy[1][i], w10) + __mul24(gy[1][i + 1], w11), 14);
        }
    }
    int pa11 = 0, pa12 = 0, pa22 = 0;
    if (active) {
#pragma unroll
        for (int i = 0; i < 7; ++i) { pa11 += __mul24(Ixv[i], Ixv[i]); pa12 += __mul24(Ixv[i], Iyv[i]); pa22 += __mul24(Iyv[i], Iyv[i]); }
    }
    const float A11 = (float)lk_wave_sum(pa11) * FLT_SCALE;
    const float A12 = (float)lk_wave_sum(pa12) * FLT_SCALE;
    const float A22 = (float)lk_wave_sum(pa22) * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
    if (minEig < 1e-4f || D < FLT_EPSILON) { if (level == 0) status = false; return; }
    D = 1.f / D;

    nextPt.x -= half; nextPt.y -= half;
    float2 prevDelta = make_float2(0.f, 0.f);
    int jx0 = 0, jy0 = 0, jax0 = 0;
    bool have_tile = false;
    auto stage_j = [&](int inx, int iny) {
        jx0 = inx - JT_MARGIN; jy0 = iny - JT_MARGIN; jax0 = jx0 & ~3;
        const long long lk_tj = LKNOW(); (void)lk_tj;
        __builtin_amdgcn_wave_barrier();
        lk_stage_tile(sJ, JT_PITCH, JT_ROWS, J, jax0, jy0, jx0, 32, lane);
        LKADD(10, 1); if (!(jx0 >= 0 && jx0 + 32 <= J.w && jy0 >= 0 && jy0 + JT_ROWS <= J.h)) LKADD(11, 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        LKADD(13, LKNOW() - lk_tj);
        have_tile = true;
    };
    auto outside_image = [&](int inx, int iny) { return inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h; };
    auto outside_tile = [&](int inx, int iny) { return inx < jx0 || inx > jx0 + 2 * JT_MARGIN || iny < jy0 || iny > jy0 + 2 * JT_MARGIN; };
    {   // the J tile of the first iterate: the speculative one if the iterate lies inside its margin, else staged now — BEFORE the next pass's tiles are
        // requested (loads complete in order: a staging behind them would wait for them)
        const int inx = (int)floorf(nextPt.x), iny = (int)floorf(nextPt.y);
        if (have_j && inx >= pjx0 && inx <= pjx0 + 2 * JT_MARGIN && iny >= pjy0 && iny <= pjy0 + 2 * JT_MARGIN) {
            jx0 = pjx0; jy0 = pjy0; jax0 = jx0 & ~3;
            lk_commit(sJ, pjv, JT_PITCH, JT_ROWS, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            have_tile = true;
            LKADD(14, 1);
        } else if (max_count > 0 && !outside_image(inx, iny)) stage_j(inx, iny);
    }
    if (nx.valid) {    // request the next pass's tiles now: they arrive while this pass iterates
        const float nscale = (float)(1. / (1 << nx.level));
        if (nx.prev_known) {
            const int nipx = (int)floorf(nx.prev.x * nscale - half), nipy = (int)floorf(nx.prev.y * nscale - half);
            if (!(nipx < -WIN || nipx >= nx.I.w || nipy < -WIN || nipy >= nx.I.h)) pre.i_ok = lk_prefetch(piv, IT_PITCH, IT_ROWS, nx.I, (nipx - 1) & ~3, nipy - 1, nipx - 1, 24, lane);
        }
        // start iterate of the next pass if this one stopped right here: mode 0 -> 2 x (nextPt + half) - half, mode 1 -> prev x scale - half
        const float gx = nx.mode == 0 ? (nextPt.x + half) * 2.f - half : nx.prev.x * nscale - half;
        const float gy = nx.mode == 0 ? (nextPt.y + half) * 2.f - half : nx.prev.y * nscale - half;
        const int gjx0 = (int)floorf(gx) - JT_MARGIN, gjy0 = (int)floorf(gy) - JT_MARGIN;
        if (!(gjx0 + JT_MARGIN < -WIN || gjx0 + JT_MARGIN >= nx.J.w || gjy0 + JT_MARGIN < -WIN || gjy0 + JT_MARGIN >= nx.J.h)) { pre.j_ok = lk_prefetch(pjv, JT_PITCH, JT_ROWS, nx.J, gjx0 & ~3, gjy0, gjx0, 32, lane); pre.jx0 = gjx0; pre.jy0 = gjy0; }
    }
    const long long lk_t1 = LKNOW(); (void)lk_t1;
    LKADD(1, 1); LKADD(5, lk_t1 - lk_t0); if (lk_slow_i) { LKADD(8, 1); LKADD(9, lk_t1 - lk_t0); }
    int lk_it = 0; (void)lk_it;
    // The iterations run in an inner loop that touches no global memory (so nothing in it waits for the tiles in flight); it is left when the iterate drifts out
    // of the staged tile, and the outer loop stages the tile around it.  Same sequence of operations per iteration as the single loop it replaces.
    int j = 0; bool done = false;
    while (!done && j < max_count) {
        {
            const int inx = (int)floorf(nextPt.x), iny = (int)floorf(nextPt.y);
            if (outside_image(inx, iny)) { if (level == 0) status = false; break; }
            if (!have_tile || outside_tile(inx, iny)) stage_j(inx, iny);
        }
        for (; j < max_count; ++j) {
            const int inx = (int)floorf(nextPt.x), iny = (int)floorf(nextPt.y);
            if (outside_image(inx, iny)) { if (level == 0) status = false; done = true; break; }
            if (outside_tile(inx, iny)) break;
            LKADD(0, 1); ++lk_it;
            lk_weights(nextPt.x - inx, nextPt.y - iny, w00, w01, w10, w11);
            int pb1 = 0, pb2 = 0;
            {
                const uint8_t* T = sJ + (iny - jy0 + r) * JT_PITCH + (inx - jax0 + c0);
                int top[8], bot[8];
                { uint64_t t8, b8; __builtin_memcpy(&t8, T, 8); __builtin_memcpy(&b8, T + JT_PITCH, 8);      // two 8-byte LDS reads instead of 16 byte reads
#pragma unroll
                  for (int i = 0; i < 8; ++i) { top[i] = (int)((t8 >> (8 * i)) & 0xff); bot[i] = (int)((b8 >> (8 * i)) & 0xff); } }
                if (active) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) {
                        int diff = lk_descale(__mul24(top[i], w00) + __mul24(top[i + 1], w01) + __mul24(bot[i], w10) + __mul24(bot[i + 1], w11), 14 - 5) - Iv[i];
                        pb1 += __mul24(diff, Ixv[i]);
                        pb2 += __mul24(diff, Iyv[i]);
                    }
                }
            }
            const float b1 = (float)lk_wave_sum(pb1) * FLT_SCALE;
            const float b2 = (float)lk_wave_sum(pb2) * FLT_SCALE;
            const float2 delta = make_float2((A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D);
            nextPt.x += delta.x; nextPt.y += delta.y;
            next = make_float2(nextPt.x + half, nextPt.y + half);
            if ((double)delta.x * (double)delta.x + (double)delta.y * (double)delta.y <= eps_sq) { done = true; break; }
            if (j > 0 && (double)fabsf(delta.x + prevDelta.x) < 0.01 && (double)fabsf(delta.y + prevDelta.y) < 0.01) {
                next.x -= delta.x * 0.5f; next.y -= delta.y * 0.5f;
                done = true; break;
            }
            prevDelta = delta;
        }
    }
    LKADD(4, LKNOW() - lk_t1); LKMAX(12, lk_it);
    if (status && level == 0) {   // err != NULL at the reference call sites: final in-bounds re-check
        const int ix = (int)floorf(next.x - half), iy = (int)floorf(next.y - half);
        if (ix < -WIN || ix >= J.w || iy < -WIN || iy >= J.h) status = false;
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool lk_in_border(float2 pt, int rows, int cols) {   // feature_utils.h:68-74
    const int x = __float2int_rn(pt.x), y = __float2int_rn(pt.y);
    return 1 <= x && x < cols - 1 && 1 <= y && y < rows - 1;
}

// generic single-direction calcOpticalFlowPyrLK
__global__ __launch_bounds__(64) void lk_generic_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, int n, int max_level,
                                                        int iters, double eps_sq, int use_initial, float2* __restrict__ pts_b,
                                                        uint8_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t sI[IT_ROWS * IT_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t sJ[JT_ROWS * JT_PITCH];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n) return;
    const float2 prev = pts_a[p];
    float2 next = use_initial ? pts_b[p] : make_float2(0.f, 0.f);
    bool st = true;
    LkPre pre; pre.i_ok = pre.j_ok = 0;
    uint32_t piv[LK_PRE_I], pjv[LK_PRE_J];
    LkNext none; none.valid = 0;
    for (int level = max_level; level >= 0; --level) {
        const DvLevel I = A.L[level], J = B.L[level];
        lk_level(I, J, level, max_level, prev, next, st, iters, eps_sq, use_initial != 0, sI, sJ, lane, pre, piv, pjv, none);
    }
    if (lane == 0) { pts_b[p] = next; status[p] = st ? 1 : 0; }
}

__device__ __forceinline__ DvLevel lk_level_of(const DvPyr& P, int l) {      // wave-uniform (the descriptors may come out of LDS: tell the compiler)
    const DvLevel v = P.L[l]; DvLevel r;
    const unsigned long long a = (unsigned long long)v.p;
    r.p = (uint8_t*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
    r.w = __builtin_amdgcn_readfirstlane(v.w); r.h = __builtin_amdgcn_readfirstlane(v.h); r.pitch = __builtin_amdgcn_readfirstlane(v.pitch); r.apron = __builtin_amdgcn_readfirstlane(v.apron);
    return r;
}
__device__ __forceinline__ DvLevel lk_pick(bool c, const DvLevel& a, const DvLevel& b) { DvLevel r; r.p = c ? a.p : b.p; r.w = c ? a.w : b.w; r.h = c ? a.h : b.h; r.pitch = c ? a.pitch : b.pitch; r.apron = c ? a.apron : b.apron; return r; }
// FeatureTrackByLK fused: fwd (maxLevel 3) + bwd (maxLevel 1, initial flow) + distance + InBorder, for the wave's point p
__device__ __forceinline__ void lk_track_point(const DvPyr& A, const DvPyr& B, const float2* __restrict__ pts_a, int p, int flow_back, float dist_thresh, double eps_sq,
                                               float2* __restrict__ pts_b, uint8_t* __restrict__ status, float add_x, float add_y, int use_add, uint8_t* sI, uint8_t* sJ, int lane) {
    const long long lk_p0 = LKNOW(); (void)lk_p0;
    float2 prev = pts_a[p];
    if (use_add) { prev.x = prev.x + add_x; prev.y = prev.y + add_y; }      // InstFeat::TrackRightByPad: ROI coordinates + box2d->rect.tl() (float + float)
    float2 next = make_float2(0.f, 0.f), rev = prev;
    bool st = true, rst = true;
    const int mlf = min(3, A.levels - 1), mlb = min(1, A.levels - 1);
    const int nf = mlf + 1, npass = nf + (flow_back ? mlb + 1 : 0);
    // ONE loop over the level passes (forward 3..0, then backward 1..0 with the forward result as the previous point and the original point as the initial
    // flow): lk_level is inlined once, and the tiles each pass requests for the next one stay in registers across the loop
    LkPre pre; pre.i_ok = pre.j_ok = 0;
    uint32_t piv[LK_PRE_I], pjv[LK_PRE_J];
#pragma unroll
    for (int u = 0; u < LK_PRE_I; ++u) piv[u] = 0u;
#pragma unroll
    for (int u = 0; u < LK_PRE_J; ++u) pjv[u] = 0u;
    for (int pass = 0; pass < npass; ++pass) {
        const bool bwd = pass >= nf;
        const int level = bwd ? mlb - (pass - nf) : mlf - pass;
        const DvLevel la = lk_level_of(A, level), lb = lk_level_of(B, level);
        const DvLevel I = lk_pick(bwd, lb, la), J = lk_pick(bwd, la, lb);
        LkNext nx; nx.valid = pass + 1 < npass;
        if (nx.valid) {
            const bool nb = pass + 1 >= nf;
            const int nl = nb ? mlb - (pass + 1 - nf) : level - 1;
            const DvLevel na = lk_level_of(A, nl), nbl = lk_level_of(B, nl);
            nx.I = lk_pick(nb, nbl, na); nx.J = lk_pick(nb, na, nbl); nx.level = nl;
            if (nb && !bwd) { nx.prev = prev; nx.prev_known = 0; nx.mode = 1; }          // forward level 0 -> first backward level: its previous point is this pass's result
            else { nx.prev = bwd ? next : prev; nx.prev_known = 1; nx.mode = 0; }
        }
        float2 cur = bwd ? rev : next; bool cst = bwd ? rst : st;
        lk_level(I, J, level, bwd ? mlb : mlf, bwd ? next : prev, cur, cst, 30, eps_sq, bwd, sI, sJ, lane, pre, piv, pjv, nx);
        if (bwd) { rev = cur; rst = cst; } else { next = cur; st = cst; }
    }
    if (flow_back) {
        const float dx = prev.x - rev.x, dy = prev.y - rev.y;
        st = st && rst && sqrtf(dx * dx + dy * dy) <= dist_thresh;
    }
    if (st && !lk_in_border(next, B.L[0].h, B.L[0].w)) st = false;
    if (lane == 0) { pts_b[p] = next; status[p] = st ? 1 : 0; }
    LKADD(6, 1); LKADD(7, LKNOW() - lk_p0); LKMAX(3, LKNOW() - lk_p0);
}
// Position order of the points of one launch (round 5): rank of every point under the key (row, column) of its pixel, ties by index — order[rank] = point.  One workgroup,
// rank by counting (n <= 1024: n LDS broadcast reads per thread).  lk_track_kernel deals the SORTED points to the XCDs in contiguous ranges (workgroup b runs on XCD b & 7):
// an XCD's waves then read one horizontal band of the two pyramids instead of all of both (points arrive in track order: by age, not by position — PMC: 5.4 MB fetched per
// launch for 1.6 MB of tiles, the pyramids pulled into up to eight L2s).  Which workgroup tracks which point changes, nothing else: results are bit-identical.
__global__ __launch_bounds__(1024) void lk_order_kernel(const float2* __restrict__ pts, const int* __restrict__ n_dev, int n_host, unsigned short* __restrict__ order) {
    __shared__ unsigned key[1024];
    const int n = min(n_dev ? *n_dev : n_host, 1024), t = threadIdx.x;
    if (t < n) {
        const float2 q = pts[t];
        const int yi = min(max((int)q.y, 0), 8191), xi = min(max((int)q.x, 0), 8191);      // (NaN / out-of-image positions: clamped — any key is a valid order)
        key[t] = ((unsigned)yi << 13) | (unsigned)xi;
    }
    __syncthreads();
    if (t < n) {
        const unsigned k = key[t];
        int r = 0;
        for (int j = 0; j < n; ++j) { const unsigned kj = key[j]; r += (kj < k || (kj == k && j < t)) ? 1 : 0; }
        order[r] = (unsigned short)t;
    }
}
__global__ __launch_bounds__(64) void lk_track_kernel(DvPyr A, DvPyr B, const float2* __restrict__ pts_a, const int* __restrict__ n_dev,
                                                      int n_host, int flow_back, float dist_thresh, double eps_sq,
                                                      float2* __restrict__ pts_b, uint8_t* __restrict__ status, float add_x, float add_y, int use_add,
                                                      const unsigned short* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) uint8_t sI[IT_ROWS * IT_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t sJ[JT_ROWS * JT_PITCH];
    const int lane = threadIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    int p = blockIdx.x;
    if (order) {           // XCD b & 7 takes the sorted positions [(b & 7) per, (b & 7) per + per): one-to-one from the padded grid onto [0, 8 per)
        const int per = (n + 7) >> 3, L = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
        if (((int)blockIdx.x >> 3) >= per || L >= n) return;
        p = order[L];
    }
    if (p >= n) return;
    // the level descriptors are indexed by the pass loop: from LDS (a dynamic index into the by-value argument structs makes the compiler copy them to scratch,
    // i.e. a memory round trip per pass)
    __shared__ DvPyr s_pyr[2];
    if (lane == 0) { s_pyr[0] = A; s_pyr[1] = B; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    lk_track_point(s_pyr[0], s_pyr[1], pts_a, p, flow_back, dist_thresh, eps_sq, pts_b, status, add_x, add_y, use_add, sI, sJ, lane);
}
// the same for several independent jobs in ONE launch (the per-object trackers of dynamic mode: blockIdx.y = job): the latency of a launch — one wave
// per point, up to 180 dependent iterations — is paid once per stage instead of once per object
__global__ __launch_bounds__(64) void lk_track_multi_kernel(const DvLkJob* __restrict__ jobs, int flow_back, float dist_thresh, double eps_sq) {
    __shared__ __attribute__((aligned(16))) uint8_t sI[IT_ROWS * IT_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t sJ[JT_ROWS * JT_PITCH];
    const DvLkJob& j = jobs[blockIdx.y];
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= *j.n_dev) return;
    lk_track_point(j.A, j.B, j.pts_a, p, flow_back, dist_thresh, eps_sq, j.pts_b, j.status, j.add_x, j.add_y, j.use_add, sI, sJ, lane);
}

void dv_launch_lk_generic(const DvPyr& A, const DvPyr& B, const float2* pts_a, int n, int max_level, int iters, double eps_sq,
                          int use_initial, float2* pts_b, uint8_t* status, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lk_generic_kernel, dim3(n), dim3(64), 0, s, A, B, pts_a, n, max_level, iters, eps_sq, use_initial, pts_b, status);
}

void dv_launch_lk_track(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max, int flow_back,
                        float dist_thresh, float2* pts_b, uint8_t* status, hipStream_t s, unsigned short* order_scratch) {
    if (n_max <= 0) return;
    const double eps = 0.01;   // TermCriteria default / feature_utils.cpp:52; OpenCV squares it
    const bool ordered = order_scratch && n_max <= 1024;          // position-sorted points, XCD-aware (lk_order_kernel)
    if (ordered) hipLaunchKernelGGL(lk_order_kernel, dim3(1), dim3(1024), 0, s, pts_a, n_dev, n_max, order_scratch);
    hipLaunchKernelGGL(lk_track_kernel, dim3(ordered ? (n_max + 7) & ~7 : n_max), dim3(64), 0, s, A, B, pts_a, n_dev, n_max, flow_back, dist_thresh, eps * eps, pts_b, status, 0.f, 0.f, 0,
                       ordered ? order_scratch : nullptr);
}
void dv_launch_lk_track_offset(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max, int flow_back,
                               float dist_thresh, float add_x, float add_y, float2* pts_b, uint8_t* status, hipStream_t s) {
    if (n_max <= 0) return;
    const double eps = 0.01;
    hipLaunchKernelGGL(lk_track_kernel, dim3(n_max), dim3(64), 0, s, A, B, pts_a, n_dev, n_max, flow_back, dist_thresh, eps * eps, pts_b, status, add_x, add_y, 1, nullptr);
}
void dv_launch_lk_track_multi(const DvLkJob* jobs_dev, int n_jobs, int n_max, int flow_back, float dist_thresh, hipStream_t s) {
    if (n_jobs <= 0 || n_max <= 0) return;
    const double eps = 0.01;
    hipLaunchKernelGGL(lk_track_multi_kernel, dim3(n_max, n_jobs), dim3(64), 0, s, jobs_dev, flow_back, dist_thresh, eps * eps);
}

// VIODE (cfg::dataset == kViode): InstFeat::TrackRightByPad keeps a right-image point only if the RIGHT camera's segmentation image carries the object's key at that
// pixel — status[i] && VIODE::PixelToKey(right_points[i], img.seg1) != id -> 0 (front_end/instance_feature.cpp:263-268; cv::Mat::at(Point2f) rounds half to even).
// One workgroup per job of the stereo LK table, behind that launch on the same stream; key_img = dv_viode_mask's key image of seg1.
__global__ __launch_bounds__(64) void right_key_check_kernel(const DvLkJob* __restrict__ jobs, const uint32_t* __restrict__ key_img, int pitch, int w, int h) {
    const DvLkJob j = jobs[blockIdx.x];
    const int n = *j.n_dev;
    for (int i = threadIdx.x; i < n; i += 64) {
        if (!j.status[i]) continue;
        const float2 p = j.pts_b[i];
        const int x = min(max(__float2int_rn(p.x), 0), w - 1), y = min(max(__float2int_rn(p.y), 0), h - 1);      // (InBorder has already put the point inside the image)
        if (key_img[(size_t)y * pitch + x] != j.key) j.status[i] = 0;
    }
}
void dv_launch_right_key_check(const DvLkJob* jobs_dev, int n_jobs, const uint32_t* key_img, int pitch_elems, int w, int h, hipStream_t s) {
    if (n_jobs > 0 && key_img) hipLaunchKernelGGL(right_key_check_kernel, dim3(n_jobs), dim3(64), 0, s, jobs_dev, key_img, pitch_elems, w, h);
}
