/*
 * dvins.h — C ABI of libdvins_hip.so: the MI355X (gfx950) implementation of dynamic_vins'
 * hot path (front-end feature tracking + sliding-window bundle adjustment).
 *
 * The reference has no plugin/FFI layer: its seam is the two C++ classes FeatureTracker
 * (dynamic_vins/src/front_end/background_tracker.h:41-88) and Estimator
 * (dynamic_vins/src/estimator/estimator.h:55-164).  The header-only C++ shims in
 * dynamic_vins_amd/host/ keep those class signatures and call the entry points below; a
 * reference maintainer swaps the bodies of the cited functions for these calls
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: every function returns 0 on success, <0 on error (message via
 * dv_last_error); no exceptions cross the ABI; the caller owns every buffer it passes;
 * the ctx owns device memory and streams; one ctx per thread.  Image/point buffers are host
 * memory when mem == DV_MEM_HOST and device (HBM) memory when mem == DV_MEM_DEVICE.
 */
#ifndef DVINS_H
#define DVINS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DV_MEM_HOST   0
#define DV_MEM_DEVICE 1

#define DV_MODE_RAW   0   /* FeatureTracker::TrackImage      (background_tracker.cpp:52-158)  */
#define DV_MODE_NAIVE 1   /* FeatureTracker::TrackImageNaive (background_tracker.cpp:400-516) */

#define DV_MAX_FEATS 1024 /* device capacity for tracked points per tracker */

typedef struct dv_ctx dv_ctx;

/* camodocal::PinholeCamera parameters (camera_models/src/camera_models/PinholeCamera.cc:292-295) */
typedef struct dv_cam { double fx, fy, cx, cy, k1, k2, p1, p2; } dv_cam;

/* fe_para (front_end/front_end_parameters.cpp:18-40) + cfg::is_stereo (utils/parameters.cpp) */
typedef struct dv_config {
    int width, height;      /* image_width / image_height */
    int max_cnt;            /* max_cnt   -> fe_para::kMaxCnt  */
    int min_dist;           /* min_dist  -> fe_para::kMinDist */
    int flow_back;          /* flow_back -> fe_para::is_flow_back */
    int stereo;             /* num_of_cam == 2 */
    dv_cam cam0, cam1;
    int device;             /* HIP device ordinal (LOCAL_RANK for one-process-per-GPU) */
    int reserved[7];
} dv_config;

/* One tracked feature = one entry of FeatureBackground::points
 * (basic/frontend_feature.h:34-44): id -> [(0, left Vec7d), (1, right Vec7d)] */
typedef struct dv_feat {
    uint32_t id;
    int32_t  track_cnt;
    int32_t  has_right;
    int32_t  pad_;
    double   left[7];       /* x_n, y_n, 1, u, v, vx, vy (background_tracker.cpp:347-355) */
    double   right[7];
} dv_feat;

/* ---- lifetime ---- */
dv_ctx*     dv_create(const dv_config* cfg);      /* NULL on failure; dv_last_error(NULL) */
void        dv_destroy(dv_ctx* ctx);
const char* dv_last_error(dv_ctx* ctx);
/* resets tracker state (ids, previous frame); Estimator::ClearState analogue for the front end */
int         dv_reset(dv_ctx* ctx);
int         dv_sync(dv_ctx* ctx);                 /* block until all work queued on the ctx is done */

/* ---- front end: whole-frame entry (replaces the body of FeatureTracker::TrackImage /
 * TrackImageNaive, background_tracker.cpp:52-158 / 400-516).  gray0/gray1: CV_8UC1 rows of
 * `stride` bytes.  mask_or_null: inv_merge_mask (0 = object) for DV_MODE_NAIVE, already eroded.
 * out: >= max_cnt rows (host memory always); *n_out = rows written. */
int dv_track_stereo(dv_ctx* ctx, const uint8_t* gray0, const uint8_t* gray1, int w, int h, int stride,
                    double t, const uint8_t* mask_or_null, int mode, int mem, dv_feat* out, int* n_out);
/* asynchronous split of the same call: enqueue the frame, later collect its output.  Lets the
 * caller overlap frame k+1's tracking with frame k's bundle adjustment (the reference does this
 * with two threads, system/main.cpp:178,394-404). */
int dv_track_stereo_enqueue(dv_ctx* ctx, const uint8_t* gray0, const uint8_t* gray1, int w, int h, int stride,
                            double t, const uint8_t* mask_or_null, int mode, int mem);
int dv_track_stereo_collect(dv_ctx* ctx, dv_feat* out, int* n_out);

/* ---- front end: operator-level entries ---- */
/* cv::calcOpticalFlowPyrLK(img_a,img_b,pts_a,pts_b,status,err,Size(21,21),max_level,
 *   TermCriteria(COUNT+EPS,iters,eps), use_initial ? OPTFLOW_USE_INITIAL_FLOW : 0)
 * call sites: front_end/feature_utils.cpp:43,50.  pts: interleaved float (x,y). */
int dv_lk(dv_ctx* ctx, const uint8_t* img_a, const uint8_t* img_b, int w, int h, int stride,
          const float* pts_a, int n, int max_level, int iters, double eps, int use_initial,
          float* pts_b, uint8_t* status, int mem);
/* FeatureTrackByLK (front_end/feature_utils.cpp:35-69): fwd LK + bwd LK + distance + InBorder */
int dv_track_by_lk(dv_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int w, int h, int stride,
                   const float* pts1, int n, int flow_back, float dist_thresh,
                   float* pts2, uint8_t* status, int mem);
/* cv::goodFeaturesToTrack(img, out, max_n, quality, min_dist, mask) call sites:
 * background_tracker.cpp:85,238; instance_feature.cpp:381; dynamic_tracker.cpp:435 */
int dv_gftt(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride,
            int max_n, double quality, double min_dist, float* out_xy, int* n_out, int mem);
/* cv::cornerMinEigenVal(img, eig, 3, 3) (inside goodFeaturesToTrack) */
int dv_min_eigen(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem);
/* cv::pyrDown (inside buildOpticalFlowPyramid); dst is ((w+1)/2) x ((h+1)/2), tightly packed */
int dv_pyr_down(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int mem);
/* cv::circle(mask, pt, radius, 0, -1) per point (background_tracker.cpp:79-80) */
int dv_circle_mask(dv_ctx* ctx, uint8_t* mask, int w, int h, int stride, const float* pts_xy, int n,
                   int radius, int mem);
/* ErodeMask / ErodeMaskGpu (front_end/feature_utils.h:130-146) */
int dv_erode(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, int k, uint8_t* dst, int mem);
/* InstFeat::UndistortedPts -> PinholeCamera::liftProjective (instance_feature.cpp:94-103) */
int dv_lift_projective(dv_ctx* ctx, const dv_cam* cam, const float* pts_xy, int n, float* out_xy, int mem);

/* ---- measurement hooks (used by bench.py; HIP-event timing on the ctx's own stream) ---- */
/* names: "pyr","lk_temporal","compact","gftt_eig","gftt_select","lk_stereo","frame" */
int dv_timing_enable(dv_ctx* ctx, int on);
int dv_timing_get(dv_ctx* ctx, const char* name, double* total_ms, long long* count);

#ifdef __cplusplus
}
#endif
#endif
