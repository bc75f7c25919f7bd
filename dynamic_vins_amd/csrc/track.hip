// track.hip — feature bookkeeping of FeatureTracker::TrackImage on the device (so a frame needs no
// host round trip between LK, corner detection and stereo matching) + small operator kernels.
//
//   track_compact_kernel : ReduceVector x4 + "++track_cnt" + SortPoints
//                          (front_end/background_tracker.cpp:61-77, feature_utils.h:77-85,
//                          feature_utils.cpp:307-328) — stable compaction by block scan, stable
//                          rank sort by track_cnt (desc).  Integer/index work: bit-exact.
//   track_finalize_kernel: InstFeat::UndistortedPts / PtsVelocity / RightUndistortedPts /
//                          RightPtsVelocity / PostProcess + FeatureTracker::SetOutputFeats
//                          (front_end/instance_feature.cpp:26-146, instance_feature.h:88-101,
//                          background_tracker.cpp:340-370).  The std::map<id,point> lookups of the
//                          reference become per-slot arrays that travel with the feature.
//   circle_mask_kernel   : cv::circle(mask, pt, r, 0, -1)      (background_tracker.cpp:79-80)
//   erode_{h,v}_kernel   : cv::erode rect k x k                 (feature_utils.h:130-146)
//   lift_kernel          : PinholeCamera::liftProjective        (PinholeCamera.cc:450-508)
#include "dv_internal.h"

__device__ void dv_lift_projective_d(const dv_cam& c, double px, double py, double& ox, double& oy) {
    const double ik11 = 1.0 / c.fx, ik13 = -c.cx / c.fx, ik22 = 1.0 / c.fy, ik23 = -c.cy / c.fy;
    const double mx_d = ik11 * px + ik13, my_d = ik22 * py + ik23;
    if (c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0) { ox = mx_d; oy = my_d; return; }   // m_noDistortion
    double mx_u = mx_d, my_u = my_d;
    for (int i = 0; i < 8; ++i) {       // recursive distortion model, n = 8
        double mx2 = mx_u * mx_u, my2 = my_u * my_u, mxy = mx_u * my_u, rho2 = mx2 + my2;
        double rad = c.k1 * rho2 + c.k2 * rho2 * rho2;
        double dx = mx_u * rad + 2.0 * c.p1 * mxy + c.p2 * (rho2 + 2.0 * mx2);
        double dy = my_u * rad + 2.0 * c.p2 * mxy + c.p1 * (rho2 + 2.0 * my2);
        mx_u = mx_d - dx; my_u = my_d - dy;
    }
    ox = mx_u; oy = my_u;
}

__device__ __forceinline__ void track_compact_body(const DvTrackState& tr, const uint8_t* __restrict__ in_mask, int mask_pitch,
                                                   int sort_by_cnt, int* n_cand, unsigned* max_ord) {
    __shared__ float2 s_cur[DV_MAX_FEATS], s_pun[DV_MAX_FEATS], s_prun[DV_MAX_FEATS];
    __shared__ uint32_t s_id[DV_MAX_FEATS];
    __shared__ int s_cnt[DV_MAX_FEATS];
    __shared__ uint8_t s_rv[DV_MAX_FEATS];
    __shared__ int s_wsum[16];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = *tr.n_feat;
    bool keep = tid < n && tr.lk_status[tid] != 0;
    float2 p = make_float2(0.f, 0.f);
    if (keep) {
        p = tr.lk_pts[tid];
        if (in_mask) {   // TrackLeftGPU: mask.at<uchar>(Point2f) == 0 -> drop (instance_feature.cpp:211-216)
            int x = __float2int_rn(p.x), y = __float2int_rn(p.y);
            if (in_mask[(size_t)y * mask_pitch + x] == 0) keep = false;
        }
    }
    // stable compaction: exclusive scan of keep flags
    const unsigned long long bal = __ballot(keep);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wsum[wv] = __popcll(bal);
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int i = 0; i < 16; ++i) { int v = s_wsum[i]; s_wsum[i] = acc; acc += v; } s_total = acc; }
    __syncthreads();
    const int total = s_total;
    if (keep) {
        const int pos = s_wsum[wv] + in_wave;
        s_cur[pos] = p;
        s_id[pos] = tr.ids[tid];
        s_cnt[pos] = tr.track_cnt[tid] + 1;
        s_pun[pos] = tr.prev_un[tid];
        s_prun[pos] = tr.prev_run[tid];
        s_rv[pos] = tr.prev_rvalid[tid];
    }
    __syncthreads();
    if (tid < total) {
        int dst = tid;
        if (sort_by_cnt) {   // stable rank sort, track_cnt descending
            const int c = s_cnt[tid];
            int rank = 0;
            for (int j = 0; j < total; ++j) { int cj = s_cnt[j]; rank += (cj > c) || (cj == c && j < tid); }
            dst = rank;
        }
        tr.curr_pts[dst] = s_cur[tid];
        tr.ids[dst] = s_id[tid];
        tr.track_cnt[dst] = s_cnt[tid];
        tr.prev_un[dst] = s_pun[tid];
        tr.prev_run[dst] = s_prun[tid];
        tr.prev_rvalid[dst] = s_rv[tid];
        tr.tracked[dst] = 1;
    }
    if (tid == 0) { *tr.n_feat = total; *tr.n_tracked = total; if (n_cand) *n_cand = 0; if (max_ord) *max_ord = 0u; }
}
__global__ __launch_bounds__(1024) void track_compact_kernel(DvTrackState tr, const uint8_t* __restrict__ in_mask, int mask_pitch,
                                                             int sort_by_cnt, int* n_cand, unsigned* max_ord) {
    track_compact_body(tr, in_mask, mask_pitch, sort_by_cnt, n_cand, max_ord);
}
// the same for several trackers in ONE launch (blockIdx.x = job)
__global__ __launch_bounds__(1024) void track_compact_multi_kernel(const DvCompactJob* __restrict__ jobs) {
    const DvCompactJob j = jobs[blockIdx.x];
    track_compact_body(j.tr, j.in_mask, j.mask_pitch, j.sort_by_cnt, j.n_cand, j.max_ord);
}

__device__ __forceinline__ void track_finalize_body(const DvTrackState& tr, const dv_cam& cam0, const dv_cam& cam1, int stereo, double dt,
                                                    dv_feat* __restrict__ out, int* __restrict__ n_out, float off_x, float off_y, int use_off,
                                                    const int* __restrict__ err_in, int* __restrict__ err_out) {
    // `out` / `n_out` / `err_out` may be PINNED HOST memory (the frame's download without copy dispatches behind the kernel): the rows leave through LDS
    // so that a wave writes 1 KB of consecutive bytes, not 64 rows' fields at a 128-byte stride
    __shared__ dv_feat s_rows[256];
    const int n = *tr.n_feat;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { *n_out = n; if (err_out) *err_out = *err_in; }
    if (i < n) {
    const float2 p = tr.curr_pts[i];
    double ux, uy;
    // objects (InstFeat::UndistortedPointsWithAddOffset, instance_feature.cpp:123-133): pt.x + box2d->rect.tl().x is a float + float sum (Box2D::rect is a cv::Rect2f)
    const float lx = use_off ? p.x + off_x : p.x, ly = use_off ? p.y + off_y : p.y;
    dv_lift_projective_d(cam0, (double)lx, (double)ly, ux, uy);
    const float2 un = make_float2((float)ux, (float)uy);
    float2 vel = make_float2(0.f, 0.f);
    if (tr.tracked[i]) {
        const float2 pu = tr.prev_un[i];
        vel.x = (float)((double)(un.x - pu.x) / dt);
        vel.y = (float)((double)(un.y - pu.y) / dt);
    }
    dv_feat f;
    f.id = tr.ids[i]; f.track_cnt = tr.track_cnt[i]; f.has_right = 0; f.pad_ = 0;
    f.left[0] = un.x; f.left[1] = un.y; f.left[2] = 1.0; f.left[3] = p.x; f.left[4] = p.y; f.left[5] = vel.x; f.left[6] = vel.y;
#pragma unroll
    for (int k = 0; k < 7; ++k) f.right[k] = 0.0;
    uint8_t rv = 0;
    float2 run = make_float2(0.f, 0.f);
    if (stereo && tr.right_status[i]) {
        const float2 rp = tr.right_pts[i];
        double rx, ry;
        dv_lift_projective_d(cam1, (double)rp.x, (double)rp.y, rx, ry);
        run = make_float2((float)rx, (float)ry);
        float2 rvel = make_float2(0.f, 0.f);
        if (tr.prev_rvalid[i]) {
            const float2 pr = tr.prev_run[i];
            rvel.x = (float)((double)(run.x - pr.x) / dt);
            rvel.y = (float)((double)(run.y - pr.y) / dt);
        }
        f.has_right = 1; rv = 1;
        f.right[0] = run.x; f.right[1] = run.y; f.right[2] = 1.0; f.right[3] = rp.x; f.right[4] = rp.y; f.right[5] = rvel.x; f.right[6] = rvel.y;
    }
    s_rows[threadIdx.x] = f;
    // PostProcess: last = curr, prev_id_pts = curr_id_pts, right_prev_id_pts = right_curr_id_pts
    tr.last_pts[i] = p;
    tr.prev_un[i] = un;
    tr.prev_run[i] = run;
    tr.prev_rvalid[i] = rv;
    }
    __syncthreads();
    const int row0 = blockIdx.x * blockDim.x, nb = min((int)blockDim.x, n - row0);
    if (nb > 0) {
        const uint4* src = reinterpret_cast<const uint4*>(s_rows);
        uint4* dst = reinterpret_cast<uint4*>(out + row0);
        for (int e = threadIdx.x; e < nb * (int)(sizeof(dv_feat) / 16); e += blockDim.x) dst[e] = src[e];
    }
}
__global__ __launch_bounds__(256) void track_finalize_kernel(DvTrackState tr, dv_cam cam0, dv_cam cam1, int stereo, double dt,
                                                             dv_feat* __restrict__ out, int* __restrict__ n_out, float off_x, float off_y, int use_off,
                                                             const int* __restrict__ err_in, int* __restrict__ err_out) {
    track_finalize_body(tr, cam0, cam1, stereo, dt, out, n_out, off_x, off_y, use_off, err_in, err_out);
}
// the same for several trackers in ONE launch (blockIdx.y = job)
__global__ __launch_bounds__(256) void track_finalize_multi_kernel(const DvFinalizeJob* __restrict__ jobs) {
    const DvFinalizeJob j = jobs[blockIdx.y];
    track_finalize_body(j.tr, j.cam0, j.cam1, j.stereo, j.dt, j.out, j.n_out, j.off_x, j.off_y, j.use_off, j.err_in, j.err_out);
}

__global__ __launch_bounds__(256) void circle_mask_kernel(uint8_t* mask, int w, int h, int pitch, const float2* __restrict__ pts, int n,
                                                          int radius, const uint8_t* __restrict__ hw) {
    const int p = blockIdx.x;
    if (p >= n) return;
    const float2 pt = pts[p];
    const int cx = __float2int_rn(pt.x), cy = __float2int_rn(pt.y);
    const int side = 2 * radius + 1;
    for (int i = threadIdx.x; i < side * side; i += blockDim.x) {
        int dy = i / side - radius, dx = i - (i / side) * side - radius;
        int y = cy + dy, x = cx + dx;
        if (y < 0 || y >= h || x < 0 || x >= w) continue;
        if (abs(dx) <= (int)hw[abs(dy)]) mask[(size_t)y * pitch + x] = 0;
    }
}

// separable erosion (min filter), out-of-image samples ignored (= +inf border)
__global__ __launch_bounds__(256) void erode_h_kernel(const uint8_t* __restrict__ src, int w, int h, int spitch, int k, uint8_t* __restrict__ dst, int dpitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int a = k / 2;
    int m = 255;
    for (int d = -a; d < k - a; ++d) { int xx = x + d; if (xx >= 0 && xx < w) m = min(m, (int)src[(size_t)y * spitch + xx]); }
    dst[(size_t)y * dpitch + x] = (uint8_t)m;
}
__global__ __launch_bounds__(256) void erode_v_kernel(const uint8_t* __restrict__ src, int w, int h, int spitch, int k, uint8_t* __restrict__ dst, int dpitch) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const int a = k / 2;
    int m = 255;
    for (int d = -a; d < k - a; ++d) { int yy = y + d; if (yy >= 0 && yy < h) m = min(m, (int)src[(size_t)yy * spitch + x]); }
    dst[(size_t)y * dpitch + x] = (uint8_t)m;
}

// the erosions of several masks in one launch each (blockIdx.z = job; the grid covers the largest mask)
__global__ __launch_bounds__(256) void erode_h_multi_kernel(const DvErodeJob* __restrict__ jobs) {
    const DvErodeJob j = jobs[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= j.w || y >= j.h) return;
    const int a = j.k / 2;
    int m = 255;
    for (int d = -a; d < j.k - a; ++d) { int xx = x + d; if (xx >= 0 && xx < j.w) m = min(m, (int)j.src[(size_t)y * j.spitch + xx]); }
    j.tmp[(size_t)y * j.tpitch + x] = (uint8_t)m;
}
__global__ __launch_bounds__(256) void erode_v_multi_kernel(const DvErodeJob* __restrict__ jobs) {
    const DvErodeJob j = jobs[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= j.w || y >= j.h) return;
    const int a = j.k / 2;
    int m = 255;
    for (int d = -a; d < j.k - a; ++d) { int yy = y + d; if (yy >= 0 && yy < j.h) m = min(m, (int)j.tmp[(size_t)yy * j.tpitch + x]); }
    j.dst[(size_t)y * j.dpitch + x] = (uint8_t)m;
}

__global__ __launch_bounds__(256) void lift_kernel(dv_cam cam, const float2* __restrict__ in, int n, double off_x, double off_y, float2* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y;
    const float fx = off_x != 0.0 ? in[i].x + (float)off_x : in[i].x, fy = off_y != 0.0 ? in[i].y + (float)off_y : in[i].y;      // float + float (Box2D::rect is a cv::Rect2f), then widened
    dv_lift_projective_d(cam, (double)fx, (double)fy, x, y);
    out[i] = make_float2((float)x, (float)y);
}

void dv_launch_compact(const DvTrackState& tr, const uint8_t* in_mask, int mask_pitch, int sort_by_cnt, int* n_cand, unsigned* max_ord, hipStream_t s) {
    hipLaunchKernelGGL(track_compact_kernel, dim3(1), dim3(1024), 0, s, tr, in_mask, mask_pitch, sort_by_cnt, n_cand, max_ord);
}
void dv_launch_compact_multi(const DvCompactJob* jobs_dev, int n_jobs, hipStream_t s) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(track_compact_multi_kernel, dim3(n_jobs), dim3(1024), 0, s, jobs_dev);
}
void dv_launch_finalize_multi(const DvFinalizeJob* jobs_dev, int n_jobs, int n_max, hipStream_t s) {
    if (n_jobs <= 0) return;
    const int blocks = (n_max + 255) / 256;
    hipLaunchKernelGGL(track_finalize_multi_kernel, dim3(blocks > 0 ? blocks : 1, n_jobs), dim3(256), 0, s, jobs_dev);
}
void dv_launch_finalize(const DvTrackState& tr, const dv_cam& cam0, const dv_cam& cam1, int stereo, double dt, int n_max, dv_feat* out, int* n_out, hipStream_t s, const int* err_in, int* err_out) {
    const int blocks = (n_max + 255) / 256;
    hipLaunchKernelGGL(track_finalize_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, tr, cam0, cam1, stereo, dt, out, n_out, 0.f, 0.f, 0, err_in, err_out);
}
void dv_launch_finalize_offset(const DvTrackState& tr, const dv_cam& cam0, const dv_cam& cam1, int stereo, double dt, int n_max, double off_x, double off_y, dv_feat* out, int* n_out, hipStream_t s) {
    const int blocks = (n_max + 255) / 256;
    hipLaunchKernelGGL(track_finalize_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, tr, cam0, cam1, stereo, dt, out, n_out, (float)off_x, (float)off_y, 1, (const int*)nullptr, (int*)nullptr);
}
void dv_launch_circle_mask(uint8_t* mask, int w, int h, int pitch, const float2* pts, int n, int radius, const uint8_t* hw, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(circle_mask_kernel, dim3(n), dim3(256), 0, s, mask, w, h, pitch, pts, n, radius, hw);
}
void dv_launch_erode(const uint8_t* src, int w, int h, int spitch, int k, uint8_t* tmp, int tpitch, uint8_t* dst, int dpitch, hipStream_t s) {
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(erode_h_kernel, grid, dim3(256), 0, s, src, w, h, spitch, k, tmp, tpitch);
    hipLaunchKernelGGL(erode_v_kernel, grid, dim3(256), 0, s, tmp, w, h, tpitch, k, dst, dpitch);
}
void dv_launch_erode_multi(const DvErodeJob* jobs_dev, int n_jobs, int w_max, int h_max, hipStream_t s) {
    if (n_jobs <= 0 || w_max <= 0 || h_max <= 0) return;
    dim3 grid((w_max + 255) / 256, h_max, n_jobs);
    hipLaunchKernelGGL(erode_h_multi_kernel, grid, dim3(256), 0, s, jobs_dev);
    hipLaunchKernelGGL(erode_v_multi_kernel, grid, dim3(256), 0, s, jobs_dev);
}
void dv_launch_lift(const dv_cam& cam, const float2* in, int n, double off_x, double off_y, float2* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lift_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cam, in, n, off_x, off_y, out);
}


// ---------------------------------------------------------------------------------------------------------------
// VIODE segmentation image -> instance masks: VIODE::SetViodeMaskSimple / BuildViodeMask
// (utils/dataset/viode_utils.cpp:21-170).  key = r*1000000 + g*1000*b  (viode_utils.h:23-26: the product, sic); a pixel
// is dynamic if its key is one of `dyn_keys` (the keys whose label index is in ViodeDynamicIndex).  One pass over the
// BGR label image: merge mask (255 = object), its inverse, optionally the key image, and per key the bounding box of its
// pixels (row_min, row_max, col_min, col_max) via workgroup-local then global integer min/max (order independent).
#define VIODE_MAX_KEYS 64
__global__ __launch_bounds__(256) void viode_mask_kernel(const uint8_t* __restrict__ seg, int w, int h, int spitch, const uint32_t* __restrict__ dyn_keys, int nkeys,
                                                         uint8_t* __restrict__ merge, uint8_t* __restrict__ inv, int mpitch, uint32_t* __restrict__ key_img, int32_t* __restrict__ boxes) {
    __shared__ uint32_t s_keys[VIODE_MAX_KEYS];
    __shared__ int s_box[VIODE_MAX_KEYS][4];
    const int t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < nkeys) { s_keys[t] = dyn_keys[t]; s_box[t][0] = 0x7fffffff; s_box[t][1] = -1; s_box[t][2] = 0x7fffffff; s_box[t][3] = -1; }
    __syncthreads();
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x < w && y < h) {
        const uint8_t* p = seg + (size_t)y * spitch + 3 * x;
        const uint32_t key = (uint32_t)p[2] * 1000000u + (uint32_t)p[1] * 1000u * (uint32_t)p[0];
        int hit = -1;
        for (int k = 0; k < nkeys; ++k) if (s_keys[k] == key) { hit = k; break; }
        merge[(size_t)y * mpitch + x] = hit >= 0 ? 255 : 0;
        inv[(size_t)y * mpitch + x] = hit >= 0 ? 0 : 255;
        if (key_img) key_img[(size_t)y * w + x] = key;
        if (hit >= 0) { atomicMin(&s_box[hit][0], y); atomicMax(&s_box[hit][1], y); atomicMin(&s_box[hit][2], x); atomicMax(&s_box[hit][3], x); }
    }
    __syncthreads();
    if (t < nkeys && s_box[t][1] >= 0) {
        atomicMin(&boxes[4 * t + 0], s_box[t][0]); atomicMax(&boxes[4 * t + 1], s_box[t][1]);
        atomicMin(&boxes[4 * t + 2], s_box[t][2]); atomicMax(&boxes[4 * t + 3], s_box[t][3]);
    }
}
void dv_launch_viode_mask(const uint8_t* seg, int w, int h, int spitch, const uint32_t* dyn_keys, int nkeys, uint8_t* merge, uint8_t* inv, int mpitch,
                          uint32_t* key_img, int32_t* boxes, hipStream_t s) {
    hipLaunchKernelGGL(viode_mask_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(64, 4), 0, s, seg, w, h, spitch, dyn_keys, nkeys, merge, inv, mpitch, key_img, boxes);
}
