// pyramid.hip — cv::pyrDown for CV_8UC1 on gfx950 (used by buildOpticalFlowPyramid inside
// cv::calcOpticalFlowPyrLK; reference call sites front_end/feature_utils.cpp:43,50).
// 5x5 separable [1 4 6 4 1], (sum+128)>>8, BORDER_REFLECT_101, dst = ((w+1)/2) x ((h+1)/2).
// Integer arithmetic -> bit-exact against the oracle.
//
// Layout: one 256-thread workgroup per 64x16 output tile.  The 131x35 source tile is staged
// in LDS with dword global loads (interior tiles) or reflect-indexed byte loads (border
// tiles); the horizontal pass goes LDS->LDS (int16), the vertical pass writes uchar4.
// The level-1 launch optionally also writes the source tile's core to a pitched copy of
// level 0, so the frame is read from HBM once for "copy + first pyrDown".
// blockIdx.z selects the image of a stereo pair (two images per launch).
#include "dv_internal.h"

#define PT_W 64
#define PT_H 16
#define PS_W (2 * PT_W + 3)   // 131 source columns
#define PS_H (2 * PT_H + 3)   // 35 source rows
#define PS_PITCH 136          // 34 dwords per LDS row (covers the 4-byte aligned span)

__device__ __forceinline__ int pyr_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// one 64 x 16 output tile (bx, by) of one image
__device__ __forceinline__ void pyr_down_tile(const uint8_t* __restrict__ src, int sw, int sh, int spitch, uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                              uint8_t* __restrict__ cpy, int cpitch, int bx, int by, int rn_even = 0) {
    __shared__ __attribute__((aligned(16))) uint8_t s_src[PS_H * PS_PITCH];
    __shared__ short s_h[PS_H][PT_W];
    const int tid = threadIdx.x;
    const int ox0 = bx * PT_W, oy0 = by * PT_H;
    const int sx0 = 2 * ox0 - 2, sy0 = 2 * oy0 - 2;
    const int ax0 = sx0 - 2;                                  // 2*ox0-4: multiple of 4
    const bool interior = (ax0 >= 0) && (ax0 + PS_PITCH <= sw) && (sy0 >= 0) && (sy0 + PS_H <= sh) && ((spitch & 3) == 0)
                          && ((reinterpret_cast<uintptr_t>(src) & 3) == 0);
    // Both staging loops request a batch into registers before the first LDS store: compiled as load -> wait -> store per trip they were 5 (interior) or 19
    // (border tile: byte-wise, reflected) dependent round trips — and the coarse levels, whose launches have too few workgroups to hide one another's latency,
    // consist almost entirely of border tiles.
    if (interior) {
        uint32_t* s32 = reinterpret_cast<uint32_t*>(s_src);
        constexpr int ND = PS_H * (PS_PITCH / 4), NT = (ND + 255) / 256;
        uint32_t v[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int i = tid + 256 * u < ND ? tid + 256 * u : tid;
            const int r = i / (PS_PITCH / 4), c = i - r * (PS_PITCH / 4);
            v[u] = *reinterpret_cast<const uint32_t*>(src + (size_t)(sy0 + r) * spitch + ax0 + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) if (tid + 256 * u < ND) s32[tid + 256 * u] = v[u];
    } else {
        for (int i0 = tid; i0 < PS_H * PS_PITCH; i0 += 8 * 256) {
            uint8_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u < PS_H * PS_PITCH ? i0 + 256 * u : i0;
                const int r = i / PS_PITCH, c = i - r * PS_PITCH;
                v[u] = src[(size_t)pyr_reflect101(sy0 + r, sh) * spitch + pyr_reflect101(ax0 + c, sw)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + 256 * u < PS_H * PS_PITCH) s_src[i0 + 256 * u] = v[u];
        }
    }
    __syncthreads();
    if (cpy) {   // level-0 copy: this tile's own 128x32 source pixels
        for (int i = tid; i < 32 * 32; i += 256) {
            int r = i >> 5, q = i & 31;
            int y = 2 * oy0 + r, x = 2 * ox0 + 4 * q;
            if (y < sh && x < sw) {
                const uint8_t* p = &s_src[(r + 2) * PS_PITCH + 4 * q + 4];
                if (x + 3 < sw) *reinterpret_cast<uchar4*>(cpy + (size_t)y * cpitch + x) = make_uchar4(p[0], p[1], p[2], p[3]);
                else for (int k = 0; x + k < sw; ++k) cpy[(size_t)y * cpitch + x + k] = p[k];
            }
        }
    }
    for (int i = tid; i < PS_H * PT_W; i += 256) {
        int r = i / PT_W, c = i - r * PT_W;
        const uint8_t* p = &s_src[r * PS_PITCH + 2 * c + 2];   // source column sx0 + 2c
        s_h[r][c] = (short)(p[0] + p[4] + 4 * (p[1] + p[3]) + 6 * p[2]);
    }
    __syncthreads();
    {
        const int r = tid >> 4, q = tid & 15;
        const int y = oy0 + r, x = ox0 + 4 * q;
        if (y < dh && x < dw) {
            uint8_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int c = 4 * q + k;
                int v = s_h[2 * r][c] + s_h[2 * r + 4][c] + 4 * (s_h[2 * r + 1][c] + s_h[2 * r + 3][c]) + 6 * s_h[2 * r + 2][c];
                if (rn_even) { int q = v >> 8; const int r = v & 255; q += (r > 128 || (r == 128 && (q & 1))) ? 1 : 0; o[k] = (uint8_t)min(q, 255); }      // cuda::pyrDown: __float2int_rn of the exact sum / 256
                else o[k] = (uint8_t)((v + 128) >> 8);
            }
            if (x + 3 < dw) *reinterpret_cast<uchar4*>(dst + (size_t)y * dpitch + x) = make_uchar4(o[0], o[1], o[2], o[3]);
            else for (int k = 0; x + k < dw; ++k) dst[(size_t)y * dpitch + x + k] = o[k];
        }
    }
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1,
                                                       int sw, int sh, int spitch,
                                                       uint8_t* __restrict__ dst0, uint8_t* __restrict__ dst1, int dw, int dh, int dpitch,
                                                       uint8_t* __restrict__ copy0, uint8_t* __restrict__ copy1, int cpitch, int rn_even, int tx_n, int ty_n, int n_img) {
    // XCD-aware tile order (round 5).  Workgroups are dealt to the 8 XCDs round robin by their flat index, so with tile = block the horizontal neighbours of a tile — which
    // share its source lines: a 132-byte tile row straddles two or three 128-byte lines — sat on eight different L2s and every line was fetched from HBM up to three times
    // (PMC: 4.3 MB per launch for 1.6 MB).  Here XCD x takes a CONTIGUOUS range of the row-major tile order (image, tile row, tile column): neighbours meet in one L2.
    // The grid is one-dimensional and padded to a multiple of 8 (8 * per blocks): flat index f -> tile (f & 7) * per + (f >> 3) is then one-to-one onto [0, 8 per).
    const int total = tx_n * ty_n * n_img, flat = blockIdx.x, per = (int)gridDim.x >> 3;
    const int L = (flat & 7) * per + (flat >> 3);
    if (L >= total) return;
    const int img = L / (tx_n * ty_n), rem = L - img * tx_n * ty_n, ty = rem / tx_n, tx = rem - ty * tx_n;
    pyr_down_tile(img ? src1 : src0, sw, sh, spitch, img ? dst1 : dst0, dw, dh, dpitch, img ? copy1 : copy0, cpitch, tx, ty, rn_even);
}
// the same level step for several independent image pairs in ONE launch (the per-object ROI pyramids of dynamic mode: blockIdx.z = 2 * job + image); the grid
// covers the largest job, tiles outside a smaller one end at once
__global__ __launch_bounds__(256) void pyr_down_multi_kernel(const DvPyrJob* __restrict__ jobs, int tx_n, int ty_n, int n_z) {
    // the XCD-aware tile order of pyr_down_kernel over (plane z = 2 * job + image, tile row, tile column): one-dimensional grid padded to a multiple of 8
    const int total = tx_n * ty_n * n_z, per = (int)gridDim.x >> 3, L = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (L >= total) return;
    const int z = L / (tx_n * ty_n), rem = L - z * tx_n * ty_n, ty = rem / tx_n, tx = rem - ty * tx_n;
    const DvPyrJob j = jobs[z >> 1];
    if (tx * PT_W >= j.dw || ty * PT_H >= j.dh) return;
    const int img = z & 1;
    if (img && !j.src1) return;
    pyr_down_tile(img ? j.src1 : j.src0, j.sw, j.sh, j.spitch, img ? j.dst1 : j.dst0, j.dw, j.dh, j.dpitch, img ? j.cpy1 : j.cpy0, j.cpitch, tx, ty);
}
void dv_launch_pyr_down_multi(const DvPyrJob* jobs_dev, int n_jobs, int max_dw, int max_dh, hipStream_t s) {
    if (n_jobs <= 0) return;
    const int tx_n = (max_dw + PT_W - 1) / PT_W, ty_n = (max_dh + PT_H - 1) / PT_H, n_z = 2 * n_jobs, per = (tx_n * ty_n * n_z + 7) / 8;
    hipLaunchKernelGGL(pyr_down_multi_kernel, dim3(8 * per), dim3(256), 0, s, jobs_dev, tx_n, ty_n, n_z);
}

// BORDER_REFLECT_101 apron of every level of one or two pyramids in ONE launch (blockIdx.y = level, blockIdx.z = image): with it no LK tile touches the
// byte-wise border path (17 - 22 % of the tiles did, and the slowest point of a launch — which sets the launch's duration — was almost always one of them)
__global__ __launch_bounds__(256) void pyr_apron_kernel(DvPyr a, DvPyr b) {
    const DvPyr& P = blockIdx.z ? b : a;
    if ((int)blockIdx.y >= P.levels) return;
    const DvLevel L = P.L[blockIdx.y];
    const int A = L.apron, w = L.w, h = L.h, W = w + 2 * A;
    if (A <= 0) return;
    const int top = A * W, mid = h * 2 * A, total = 2 * top + mid;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < total; k += gridDim.x * 256) {
        int px, py;
        if (k < top) { py = k / W; px = k - py * W; }
        else if (k < top + mid) { const int k2 = k - top, r = k2 / (2 * A), c = k2 - r * 2 * A; py = A + r; px = c < A ? c : w + c; }
        else { const int k3 = k - top - mid, r = k3 / W; py = A + h + r; px = k3 - r * W; }
        const int x = px - A, y = py - A;
        L.p[(ptrdiff_t)y * L.pitch + x] = L.p[(ptrdiff_t)pyr_reflect101(y, h) * L.pitch + pyr_reflect101(x, w)];
    }
}
// the aprons of several pyramids in ONE launch (the front ends of a dv_batch group: blockIdx.z = pyramid)
__global__ __launch_bounds__(256) void pyr_apron_multi_kernel(const DvPyr* __restrict__ pyrs) {
    const DvPyr& P = pyrs[blockIdx.z];
    if ((int)blockIdx.y >= P.levels) return;
    const DvLevel L = P.L[blockIdx.y];
    const int A = L.apron, w = L.w, h = L.h, W = w + 2 * A;
    if (A <= 0) return;
    const int top = A * W, mid = h * 2 * A, total = 2 * top + mid;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < total; k += gridDim.x * 256) {
        int px, py;
        if (k < top) { py = k / W; px = k - py * W; }
        else if (k < top + mid) { const int k2 = k - top, r = k2 / (2 * A), c = k2 - r * 2 * A; py = A + r; px = c < A ? c : w + c; }
        else { const int k3 = k - top - mid, r = k3 / W; py = A + h + r; px = k3 - r * W; }
        const int x = px - A, y = py - A;
        L.p[(ptrdiff_t)y * L.pitch + x] = L.p[(ptrdiff_t)pyr_reflect101(y, h) * L.pitch + pyr_reflect101(x, w)];
    }
}
void dv_launch_pyr_apron_multi(const DvPyr* pyrs_dev, int n_pyr, int max_levels, hipStream_t s) {
    if (n_pyr <= 0 || max_levels <= 0) return;
    hipLaunchKernelGGL(pyr_apron_multi_kernel, dim3(48, max_levels, n_pyr), dim3(256), 0, s, pyrs_dev);
}
void dv_launch_pyr_apron(const DvPyr& a, const DvPyr* b, hipStream_t s) {
    if (a.levels <= 0 || a.L[0].apron <= 0) return;
    hipLaunchKernelGGL(pyr_apron_kernel, dim3(48, a.levels, b ? 2 : 1), dim3(256), 0, s, a, b ? *b : a);
}

void dv_launch_pyr_down2(const uint8_t* src0, const uint8_t* src1, int sw, int sh, int spitch, uint8_t* dst0, uint8_t* dst1,
                         int dpitch, uint8_t* copy0, uint8_t* copy1, int cpitch, hipStream_t s, int rn_even) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    const int tx_n = (dw + PT_W - 1) / PT_W, ty_n = (dh + PT_H - 1) / PT_H, n_img = src1 ? 2 : 1, per = (tx_n * ty_n * n_img + 7) / 8;
    hipLaunchKernelGGL(pyr_down_kernel, dim3(8 * per), dim3(256), 0, s, src0, src1, sw, sh, spitch, dst0, dst1, dw, dh, dpitch, copy0, copy1, cpitch, rn_even, tx_n, ty_n, n_img);
}


// ---------------------------------------------------------------------------------------------------------------
// cv::cvtColor(BGR -> GRAY), 8-bit (SemanticImage::SetGrayImage[Gpu], basic/semantic_image.cpp:95-117): the 14-bit
// fixed-point weights (B 1868, G 9617, R 4899, +8192 >> 14).  Written straight into level 0 of the pyramid, so a
// colour frame is read once (3P) and never exists as a separate gray image.
// One thread = 4 pixels: three coalesced dword loads, one dword store.  blockIdx.z selects the image of a stereo pair.
__global__ __launch_bounds__(256) void bgr2gray_kernel(const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1, int w, int h, int spitch,
                                                       uint8_t* __restrict__ dst0, uint8_t* __restrict__ dst1, int dpitch) {
    const uint8_t* src = blockIdx.z ? src1 : src0;
    uint8_t* dst = blockIdx.z ? dst1 : dst0;
    const int y = blockIdx.y, x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w) return;
    const uint8_t* row = src + (size_t)y * spitch + (size_t)x4 * 3;
    uint8_t px[12];
    if (x4 + 4 <= w && ((reinterpret_cast<uintptr_t>(row) & 3) == 0)) {
        const uint32_t* r32 = reinterpret_cast<const uint32_t*>(row);
        const uint32_t a = r32[0], b = r32[1], c = r32[2];
        px[0] = a; px[1] = a >> 8; px[2] = a >> 16; px[3] = a >> 24; px[4] = b; px[5] = b >> 8; px[6] = b >> 16; px[7] = b >> 24;
        px[8] = c; px[9] = c >> 8; px[10] = c >> 16; px[11] = c >> 24;
    } else {
        for (int k = 0; k < 12; ++k) px[k] = (x4 * 3 + k < w * 3) ? row[k] : 0;
    }
    uint8_t g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = (uint8_t)((px[3 * k] * 1868 + px[3 * k + 1] * 9617 + px[3 * k + 2] * 4899 + (1 << 13)) >> 14);
    uint8_t* o = dst + (size_t)y * dpitch + x4;
    if (x4 + 4 <= w) *reinterpret_cast<uint32_t*>(o) = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
    else for (int k = 0; k < 4 && x4 + k < w; ++k) o[k] = g[k];
}

void dv_launch_bgr2gray(const uint8_t* src0, const uint8_t* src1, int w, int h, int spitch, uint8_t* dst0, uint8_t* dst1, int dpitch, hipStream_t s) {
    dim3 grid((w + 1023) / 1024, h, src1 ? 2 : 1);
    hipLaunchKernelGGL(bgr2gray_kernel, grid, dim3(256), 0, s, src0, src1, w, h, spitch, dst0, dst1, dpitch);
}

// ---------------------------------------------------------------------------------------------------------------
// cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with the fixed-point maps of cv::initUndistortRectifyMap(..., CV_16SC2)
// (cfg::is_undistort_input: ImageProcessor::Run, image_process/image_process.cpp:109-121; the merged mask,
// basic/semantic_image.cpp:86-89).  map1 = (sx, sy) int16 pairs, map2 = (fy << 5 | fx), 5 fractional bits.
// OpenCV's table weights are saturate_cast<short>(wy wx 2^15) repaired to sum 2^15; for INTER_BITS = 5 they are the
// exact integers (32 - fx)(32 - fy) 32, ... except the (0,0) cell {32767, 0, 0, 1}, whose output
// (32767 v00 + v11 + 2^14) >> 15 equals v00 = (32768 v00 + 2^14) >> 15 for 8-bit data — so the closed form below is
// bit-identical to the table.  out = (sum w v + 2^14) >> 15; neighbours outside the source read 0.
// TO_GRAY (CN == 3): the three remapped 8-bit channels go through cvtColor's 14-bit weights in registers and only the
// gray byte is written — straight into pyramid level 0: 6 B of map + 3 B of colour read and 1 B
// written per pixel instead of remap (6 + 3 + 3) + cvtColor (3 + 1).
// One thread = 4 destination pixels of a row; blockIdx.z selects the image (and its maps) of a stereo pair.
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
template <int CN, bool TO_GRAY>
__global__ __launch_bounds__(256) void remap_kernel(const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1, int w, int h, int spitch,
                                                    const short2* __restrict__ m1_0, const uint16_t* __restrict__ m2_0,
                                                    const short2* __restrict__ m1_1, const uint16_t* __restrict__ m2_1,
                                                    uint8_t* __restrict__ dst0, uint8_t* __restrict__ dst1, int dpitch) {
    const uint8_t* src = blockIdx.z ? src1 : src0;
    const short2* m1 = blockIdx.z ? m1_1 : m1_0; const uint16_t* m2 = blockIdx.z ? m2_1 : m2_0;
    uint8_t* dst = blockIdx.z ? dst1 : dst0;
    const int y = blockIdx.y, x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x4 >= w) return;
    constexpr int OC = TO_GRAY ? 1 : CN;
    uint8_t o[4 * OC];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = min(x4 + k, w - 1);
        const short2 xy = m1[(size_t)y * w + x];
        const int f = m2[(size_t)y * w + x] & 1023, fx = f & 31, fy = f >> 5;
        const int sx = xy.x, sy = xy.y;
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int v[CN];
        if ((unsigned)sx < (unsigned)(w - 1) && (unsigned)sy < (unsigned)(h - 1)) {
            const uint8_t* p = src + (size_t)sy * spitch + (size_t)sx * CN;
            if (CN == 3 && !(sy == h - 2 && sx >= w - 3)) {
                // both pixels of a row in one unaligned 8-byte load (gfx950 global loads need no alignment); 2 bytes of over-read, never
                // past the last row's end (excluded above)
                const uint64_t r0 = *reinterpret_cast<const u64_unaligned*>(p), r1 = *reinterpret_cast<const u64_unaligned*>(p + spitch);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    v[c] = (int)((r0 >> (8 * c)) & 255) * w00 + (int)((r0 >> (8 * (3 + c))) & 255) * w01 + (int)((r1 >> (8 * c)) & 255) * w10 + (int)((r1 >> (8 * (3 + c))) & 255) * w11;
            } else {
#pragma unroll
                for (int c = 0; c < CN; ++c) v[c] = p[c] * w00 + p[CN + c] * w01 + p[spitch + c] * w10 + p[spitch + CN + c] * w11;
            }
        } else {
            const bool x0 = (unsigned)sx < (unsigned)w, x1 = (unsigned)(sx + 1) < (unsigned)w, y0 = (unsigned)sy < (unsigned)h, y1 = (unsigned)(sy + 1) < (unsigned)h;
#pragma unroll
            for (int c = 0; c < CN; ++c) {
                const int a = (x0 && y0) ? src[(size_t)sy * spitch + (size_t)sx * CN + c] : 0, b = (x1 && y0) ? src[(size_t)sy * spitch + (size_t)(sx + 1) * CN + c] : 0;
                const int d = (x0 && y1) ? src[(size_t)(sy + 1) * spitch + (size_t)sx * CN + c] : 0, e = (x1 && y1) ? src[(size_t)(sy + 1) * spitch + (size_t)(sx + 1) * CN + c] : 0;
                v[c] = a * w00 + b * w01 + d * w10 + e * w11;
            }
        }
#pragma unroll
        for (int c = 0; c < CN; ++c) v[c] = (v[c] + (1 << 14)) >> 15;          // <= 255: the weights sum to 2^15
        if (TO_GRAY) o[k] = (uint8_t)((v[0] * 1868 + v[1] * 9617 + v[2] * 4899 + (1 << 13)) >> 14);
        else {
#pragma unroll
            for (int c = 0; c < CN; ++c) o[k * OC + c] = (uint8_t)v[c];
        }
    }
    uint8_t* out = dst + (size_t)y * dpitch + (size_t)x4 * OC;
    if (OC == 1 && x4 + 4 <= w) *reinterpret_cast<uint32_t*>(out) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    else for (int k = 0; k < 4 * OC && x4 * OC + k < w * OC; ++k) out[k] = o[k];
}

// cn: channels of src; to_gray (cn == 3 only): write the gray image.  dpitch in bytes, 4-byte aligned rows when the output has one channel.
void dv_launch_remap(const uint8_t* src0, const uint8_t* src1, int w, int h, int spitch, int cn, int to_gray, const int16_t* m1_0, const uint16_t* m2_0,
                     const int16_t* m1_1, const uint16_t* m2_1, uint8_t* dst0, uint8_t* dst1, int dpitch, hipStream_t s) {
    dim3 grid((w + 1023) / 1024, h, src1 ? 2 : 1);
    const short2* a = (const short2*)m1_0; const short2* b = (const short2*)m1_1;
    if (cn == 1) hipLaunchKernelGGL((remap_kernel<1, false>), grid, dim3(256), 0, s, src0, src1, w, h, spitch, a, m2_0, b, m2_1, dst0, dst1, dpitch);
    else if (to_gray) hipLaunchKernelGGL((remap_kernel<3, true>), grid, dim3(256), 0, s, src0, src1, w, h, spitch, a, m2_0, b, m2_1, dst0, dst1, dpitch);
    else hipLaunchKernelGGL((remap_kernel<3, false>), grid, dim3(256), 0, s, src0, src1, w, h, spitch, a, m2_0, b, m2_1, dst0, dst1, dpitch);
}
