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
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DV_MEM_HOST   0
#define DV_MEM_DEVICE 1
/* dv_track_stereo* / dv_batch_track_enqueue / dv_seq_input::mem / dv_seq_dynamic::mask_mem: host memory the CALLER has pinned and mapped for the device
 * (hipHostMalloc, or hipHostRegister with hipHostRegisterMapped).  The kernels read it over PCIe directly — no staging copy, no copy engine in the per-frame
 * path; the buffers must stay unchanged until the frame is collected, as device buffers must. */
#define DV_MEM_PINNED 2
/* pinned + device-mapped host memory for DV_MEM_PINNED buffers, for hosts that do not link the HIP runtime themselves (hipHostMalloc / hipHostFree); NULL on failure */
void* dv_pinned_alloc(size_t bytes);
void dv_pinned_free(void* p);
/* OR into `mem` of dv_track_stereo*: gray0 / gray1 point to 8-bit BGR frames (stride in bytes, >= 3 w); they are converted with
 * cv::cvtColor's fixed-point weights straight into pyramid level 0 (SemanticImage::SetGrayImageGpu, basic/semantic_image.cpp:103-118).
 * A mask, if given, is single-channel with stride w. */
#define DV_FMT_BGR    0x100

#define DV_MODE_RAW   0   /* FeatureTracker::TrackImage      (background_tracker.cpp:52-158)  */
#define DV_MODE_NAIVE 1   /* FeatureTracker::TrackImageNaive (background_tracker.cpp:400-516): both trackings by FeatureTrackByLKGpu */
#define DV_MODE_SEMANTIC 2 /* FeatureTracker::TrackSemanticImage (background_tracker.cpp:757-837): the background half of dynamic mode —
                             temporal tracking by FeatureTrackByLK (dist <= 0.5), right image by FeatureTrackByLKGpu (the GPU tracker, dist <= 1.0) */

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
    int mask_morphology_size; /* > 0: the inverse instance mask is eroded by a k x k rectangle first (use_mask_morphology /
                                 mask_morphology_size, background_tracker.cpp:408-416,764-768); naive / semantic modes only */
    int reserved[6];
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
struct dv_inst_det;
/* para::is_static_inst_as_background (default true, estimator/vio_parameters.h:86) — FeatureTrack, system/main.cpp:217-245: before TrackSemanticImage the pixels of
 * every visible instance the estimator reported static (dv_est_get_static_instances) are taken OUT of the merged instance mask, so the background tracker may pick
 * features on parked objects: merge_mask(row + rect.y, col + rect.x) = 0 where the instance's ROI mask is set, inv_merge_mask = ~merge_mask.
 * For the NEXT dv_track_stereo_enqueue (which must carry a mask): dets = the frame's detections (rect + ROI mask, host memory, valid until that call), static_ids =
 * the estimator's list.  Detections whose track_id is not in the list are left alone; n_static 0 cancels.  The caller's mask is not written to. */
int dv_track_unmask_static(dv_ctx* ctx, const struct dv_inst_det* dets, int n_dets, const uint32_t* static_ids, int n_static);

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
/* The reference's GPU tracker, used by dv_track_stereo* where the reference uses it (DV_MODE_NAIVE: temporal + right image; DV_MODE_SEMANTIC: right image):
 * cv::cuda::SparsePyrLKOpticalFlow::create(Size(21, 21), 3, 30[, useInitialFlow])->calc (front_end/background_tracker.cpp:34-36) — float patches sampled bilinearly,
 * Scharr derivatives on the fly, no minimum-eigenvalue test, its own pyramid (cuda::pyrDown) — and FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163: forward,
 * backward from the previous points, |p - p_rev| <= 1.0, InBorder).  Operator forms; the ctx must not be tracking a sequence. */
int dv_lk_cuda(dv_ctx* ctx, const uint8_t* img_a, const uint8_t* img_b, int w, int h, int stride, const float* pts_a, int n, int max_level, int iters, int use_initial,
               float* pts_b, uint8_t* status, int mem);
int dv_track_by_lk_gpu(dv_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int w, int h, int stride, const float* pts1, int n, int flow_back,
                       float* pts2, uint8_t* status, int mem);
int dv_pyr_down_cuda(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int mem);      /* cuda::pyrDown, 8-bit: the same sums as cv::pyrDown, rounded half to even */
/* cv::goodFeaturesToTrack(img, out, max_n, quality, min_dist, mask) call sites:
 * background_tracker.cpp:85,238; instance_feature.cpp:381; dynamic_tracker.cpp:435 */
int dv_gftt(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride,
            int max_n, double quality, double min_dist, float* out_xy, int* n_out, int mem);
/* cv::cornerMinEigenVal(img, eig, 3, 3) (inside goodFeaturesToTrack) */
int dv_min_eigen(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem);
/* The reference's GPU corner detector, used by dv_track_stereo* where the reference uses it (DV_MODE_NAIVE only: DetectNewFeature(img, use_gpu = true, ...),
 * front_end/background_tracker.cpp:445 -> front_end/instance_feature.cpp:372-379 -> DetectShiTomasiCornersGpu, front_end/feature_utils.cpp:339-348):
 * cv::cuda::createGoodFeaturesToTrackDetector(CV_8UC1, max_n, quality, min_dist)->detect(img, out, mask).  Against dv_gftt: the response map is the GPU module's
 * (float multiply-add chains, block sums inside the eigenvalue kernel), the quality threshold comes from the maximum over the WHOLE image (the CPU detector takes
 * it under the mask), candidates are eig > threshold && eig == max of the raw 3 x 3 neighbourhood; the ordering and the minimum-distance grid are the same.
 * dv_min_eigen_cuda: cv::cuda::createMinEigenValCorner(CV_8UC1, 3, 3)->compute.  Operator forms. */
int dv_gftt_cuda(dv_ctx* ctx, const uint8_t* img, const uint8_t* mask_or_null, int w, int h, int stride,
                 int max_n, double quality, double min_dist, float* out_xy, int* n_out, int mem);
int dv_min_eigen_cuda(dv_ctx* ctx, const uint8_t* img, int w, int h, int stride, float* eig, int mem);
/* cv::pyrDown (inside buildOpticalFlowPyramid); dst is ((w+1)/2) x ((h+1)/2), tightly packed */
int dv_pyr_down(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, uint8_t* dst, int mem);
/* VIODE segmentation image -> instance masks (VIODE::SetViodeMaskSimple / BuildViodeMask, utils/dataset/viode_utils.cpp:21-170).
 * key(pixel) = r*1000000 + g*1000*b (viode_utils.h:23-26, sic); a pixel is an object pixel if its key is in dyn_keys[nkeys <= 64].
 * merge_mask: 255 = object; inv_merge_mask: its complement (what TrackImageNaive / TrackSemanticImage take); key_image (optional,
 * w*h uint32): the key of every pixel (PixelToKey lookups of TrackRightByPad); boxes[nkeys][4] = row_min,row_max,col_min,col_max of
 * each key's pixels, -1 if absent (InstanceSimple).  Host buffers, tightly packed outputs. */
int dv_viode_mask(dv_ctx* ctx, const uint8_t* seg_bgr, int w, int h, int stride, const uint32_t* dyn_keys, int nkeys,
                  uint8_t* merge_mask, uint8_t* inv_merge_mask, uint32_t* key_image, int32_t* boxes);
/* cv::cvtColor(BGR2GRAY) on 8-bit images: (B 1868 + G 9617 + R 4899 + 8192) >> 14; gray is w x h, tightly packed */
int dv_bgr2gray(dv_ctx* ctx, const uint8_t* bgr, int w, int h, int stride, uint8_t* gray, int mem);
/* cv::remap(src, dst, map1, map2, INTER_LINEAR) — BORDER_CONSTANT 0 — with the fixed-point maps cv::initUndistortRectifyMap(..., CV_16SC2, ...)
 * returns (utils/camera_model.cpp:481-499): map1_xy = w*h (x, y) int16 pairs, map2 = w*h uint16 (fy << 5 | fx).  8-bit source with
 * channels 1 or 3 (stride in bytes); dst is w x h x channels, tightly packed.  The calls of ImageProcessor::Run on the colour frames
 * (image_process/image_process.cpp:109-121) and of SemanticImage::SetMask on the merged mask (basic/semantic_image.cpp:86-89).
 * The maps are host memory; src / dst follow `mem`. */
int dv_remap(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, int channels, const int16_t* map1_xy, const uint16_t* map2, uint8_t* dst, int mem);
/* cfg::is_undistort_input (utils/camera_model.cpp:481-499): installs (map1_xy != NULL) or removes (NULL) the undistortion maps of camera
 * `cam` (0 left, 1 right) in HBM.  While installed, dv_track_stereo* take DISTORTED frames and undistort them on the way into pyramid
 * level 0: with DV_FMT_BGR the remap of the three channels and cvtColor are fused (the remapped colour image never exists); gray frames
 * are remapped as one channel, which is what the reference's mono -> BGR -> remap -> gray chain yields.  w, h must equal the config's.
 * dv_config.cam0 / cam1 should then be the NEW intrinsics without distortion, as the reference resets them (camera_model.cpp:486-503).
 * A mask passed to the tracker is used as given (the reference remaps it before inverting it: use dv_remap with channels = 1). */
int dv_set_undistort_maps(dv_ctx* ctx, int cam, const int16_t* map1_xy, const uint16_t* map2, int w, int h);
/* cv::circle(mask, pt, radius, 0, -1) per point (background_tracker.cpp:79-80) */
int dv_circle_mask(dv_ctx* ctx, uint8_t* mask, int w, int h, int stride, const float* pts_xy, int n,
                   int radius, int mem);
/* ErodeMask / ErodeMaskGpu (front_end/feature_utils.h:130-146) */
int dv_erode(dv_ctx* ctx, const uint8_t* src, int w, int h, int stride, int k, uint8_t* dst, int mem);
/* InstFeat::UndistortedPts -> PinholeCamera::liftProjective (instance_feature.cpp:94-103) */
int dv_lift_projective(dv_ctx* ctx, const dv_cam* cam, const float* pts_xy, int n, float* out_xy, int mem);
/* InstFeat::UndistortedPointsWithAddOffset (front_end/instance_feature.cpp:123-133): points in ROI coordinates, the 2-D box's
 * top-left corner added in double before lifting */
int dv_lift_projective_offset(dv_ctx* ctx, const dv_cam* cam, const float* pts_xy, int n, double off_x, double off_y, float* out_xy, int mem);

/* ======================= back end: sliding-window bundle adjustment ======================= */

/* One reprojection residual block = the constructor arguments of ProjectionTwoFrameOneCamFactor (kind 0),
 * ProjectionTwoFrameTwoCamFactor (kind 1) or ProjectionOneFrameTwoCamFactor (kind 2)
 * (estimator/factor/projection_*_factor.h; built at estimator/estimator.cpp:134-178).  112 bytes. */
typedef struct dv_ba_factor {
    double pix, piy, pjx, pjy;      /* pts_i, pts_j on the normalised plane */
    double vix, viy, vjx, vjy;      /* velocity_i, velocity_j */
    double td_i, td_j;
    int32_t kind, lm, fi, fj;       /* landmark index, anchor frame (start_frame), observing frame */
    double pad_[2];
} dv_ba_factor;

/* per landmark (>= 4 observations, estimator.cpp:134-136): its factors are contiguous */
typedef struct dv_ba_lm { int32_t first, count, anchor, mask; } dv_ba_lm;

/* IMUFactor between frames fi and fj = fi+1: the IntegrationBase results (imu/integration_base.h) */
typedef struct dv_ba_imu {
    double sum_dt, dp[3], dq[4] /* w x y z */, dv[3], lin_ba[3], lin_bg[3];
    double jacobian[225], covariance[225];     /* 15x15 row-major, order P R V BA BG */
    int32_t fi, fj, pad0, pad1;
} dv_ba_imu;

/* marginalization prior in information form: cost(dx) = c0/2 + b.dx + dx.A dx/2, dx w.r.t. the linearisation
 * point x0 of the kept blocks (equivalent to MarginalizationFactor's r0 + J0 dx with A = J0^T J0, b = J0^T r0,
 * c0 = r0^T r0; factor/marginalization_factor.cpp:283-309,350-396) */
typedef struct dv_ba_prior_block { int32_t type /* 0 pose, 1 speed-bias, 2 ex_pose, 3 td */, idx, off, size_local; } dv_ba_prior_block;
typedef struct dv_ba_prior {
    int32_t valid, n, nblocks, pad;
    double c0;
    dv_ba_prior_block blocks[16];
    double x0[16][9];
} dv_ba_prior;

typedef struct dv_ba_problem {
    int32_t nframes, nlm, nfac, nimu;     /* frames in the window (frame+1), landmarks, residual blocks, IMU factors */
    int32_t use_imu, plane_kind /* 0 none, 1 PoseConstraint with IMU (dz=0), 2 vision-only (dy=0) */, max_iters;
    int32_t free_blocks; /* bit 0: para_ex_pose[0..1] are NOT SetParameterBlockConstant (estimate_extrinsic 1 once openExEstimation is set), bit 1: para_td is not
                            (estimate_td 1 while |Vs[0]| >= 0.2) — Estimator::AddBodyParameterBlock, estimator/estimator.cpp:87-100.  0 in every shipped configuration.
                            Their columns follow the frames' in the reduced system (6 + 6 + 1); such a solve takes the generic factorisation and two more launches
                            per linearisation (csrc/be_ext.hip); not available on a landmark-sharded window */
    double g_norm;
    double* pose;        /* [nframes][7] x y z qx qy qz qw   (para_pose, in/out) */
    double* speed_bias;  /* [nframes][9]                      (para_speed_bias, in/out; ignored if !use_imu) */
    double* ex_pose;     /* [2][7]                            (para_ex_pose; in/out when free_blocks bit 0 is set, else constant) */
    double* td;          /* [1]                               (para_td; in/out when free_blocks bit 1 is set, else constant) */
    double* inv_depth;   /* [nlm]                             (para_point_features, in/out) */
    const dv_ba_factor* factors; const dv_ba_lm* landmarks; const dv_ba_imu* imu;
    const dv_ba_prior* prior; const double* prior_A; const double* prior_b;     /* prior may be NULL */
    double x_norm2_extra;   /* squared norm of further free parameter blocks that count in ceres' parameter-tolerance test |step| <= 1e-8 (|x| + 1e-8) without moving:
                               the lineProjectionFactor blocks of AddLineResidualBlock under the reference's zero sqrt_info; 0 otherwise */
} dv_ba_problem;

typedef struct dv_ba_summary {
    int32_t iterations, successful, termination /* 0 max iterations, 1 converged, 2 failure */, slots;
    double initial_cost, final_cost;
} dv_ba_summary;

/* Replaces ceres::Solve in Estimator::Optimization (estimator/estimator.cpp:261-326): DENSE_SCHUR + DOGLEG,
 * HuberLoss(1.0) on reprojection blocks, max_num_iterations = max_iters, wall-clock budget disabled.
 * The whole trust-region loop runs on the device; states are updated in place. */
int dv_ba_solve(dv_ctx* ctx, dv_ba_problem* problem, dv_ba_summary* summary);

/* One evaluation at the given states (what one ceres Evaluate() pass + the Schur elimination produce): total cost, the reduced
 * camera system S = H_pp - sum_l w_l w_l^T / h_l (n x n, row-major, symmetric) and its right-hand side g = g_p - sum_l w_l g_l / h_l.
 * Column order: per frame, 6 pose columns (if the pose is free) then 9 speed-bias columns (if use_imu); then, if free (free_blocks), 6 + 6 extrinsic columns and the td
 * column; *n = their count (<= 178).
 * The sum over landmarks, IMU factors and the prior is linear, so a window sharded BY LANDMARK over several GPUs (IMU factors
 * and prior on one rank) is assembled by adding the per-rank [S | g | cost] — the all-reduce of SURVEY 8(e)
 * (dynamic_vins_amd/dist.py:allreduce_reduced_system). cost, S, g may be NULL. */
int dv_ba_eval(dv_ctx* ctx, const dv_ba_problem* problem, int* n, double* cost, double* S, double* g);

/* ---- one window sharded BY LANDMARK over several GPUs (BASELINE north_star: "RCCL all-reduce of the reduced camera-pose Hessian") ----
 * The reference has no multi-GPU path; this is the exchange step its Schur elimination (ceres DENSE_SCHUR inside Estimator::Optimization,
 * estimator.cpp:296-326) admits: rank r owns a contiguous range of landmarks, evaluates and eliminates only those, and the per-rank partial reduced
 * systems are summed IN RANK ORDER on every rank (identical bits everywhere).  After dv_dist_init_*, dv_ba_solve / dv_ba_eval / dv_est_process on this
 * ctx run sharded: every rank must be given the SAME problem and makes the same calls; every rank returns the same result.
 * A binding in the reference would sit in Estimator::SetParameter (one ctx per rank, rank / world from the launcher). */
typedef int (*dv_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);      /* recv: world x bytes_per_rank, rank-major; 0 = ok */
int dv_dist_unique_id(uint8_t id[128]);                                                  /* ncclGetUniqueId: rank 0 creates it, the launcher broadcasts it */
int dv_dist_init_rccl(dv_ctx* ctx, int rank, int world, const uint8_t id[128]);          /* RCCL all-gather on the BA stream (xGMI); no host round trip per iteration */
int dv_dist_init_host(dv_ctx* ctx, int rank, int world, dv_allgather_fn fn, void* user); /* exchange staged through pinned host memory and the caller's all-gather */
/* one-shot exchange: every rank writes its vector into a window each peer exposes through hipIpc (direct xGMI link), then a sequence flag; no
 * library call per exchange.  prepare -> all-gather the 64-byte handles with the launcher's own means -> init. */
int dv_dist_peer_prepare(dv_ctx* ctx, int rank, int world, uint8_t handle[64]);
int dv_dist_init_peer(dv_ctx* ctx, const uint8_t* handles /* [world][64], rank order */);
int dv_dist_rccl_ranks(dv_ctx* ctx, int* n);                                             /* ncclCommCount of the communicator behind dv_dist_init_rccl (0: other transport) */
int dv_dist_shutdown(dv_ctx* ctx);
/* what one rank contributes to the exchanges of a sharded window solve (bytes): `system` once per linearisation — the partial reduced camera system and the coefficients of the
 * quadratic forms the trust-region step needs of the landmarks it does not own: 12 424 doubles whatever the window holds —, `cost` once per cost-only slot, `depth` once per solve */
int dv_dist_exchange_bytes(dv_ctx* ctx, int n_landmarks, long long* system_bytes, long long* cost_bytes, long long* depth_bytes);
int dv_dist_info(dv_ctx* ctx, int* rank, int* world, int* transport, long long* exchanges);
/* operator form of the exchange: S_g (host, n doubles — this rank's partial [S | g | cost], e.g. of dv_ba_eval on the rank's share of the landmarks)
 * is replaced by the rank-ordered sum over all ranks */
int dv_allreduce_reduced_system(dv_ctx* ctx, double* S_g, int n);

/* ---- several independent sequences on ONE GPU whose window solves share every launch ("batched") ----
 * No counterpart in the reference (one process per sequence).  The members are ordinary contexts with their own estimators; after
 * dv_est_process_begin (or _dynamic_begin) has been called on every member that has a frame, dv_batch_enqueue launches the iteration slots of all
 * pending window solves as one launch per stage (argument tables in HBM, window index in the grid); dv_est_process_end then collects each member
 * as usual.  A member collected without dv_batch_enqueue is solved on its own stream.  Results are bit-identical to the unbatched path. */
typedef struct dv_batch dv_batch;
dv_batch* dv_batch_create(dv_ctx* const* ctxs, int n);      /* idle contexts of one device; NULL + dv_last_error(NULL) on failure */
void dv_batch_destroy(dv_batch* batch);                      /* the members stay valid; dv_destroy of a member before the batch detaches it from the batch */
int dv_batch_enqueue(dv_batch* batch);
int dv_batch_arrive(dv_batch* batch);                        /* one host thread per member: blocks until every member's thread has arrived; the last one enqueues */
int dv_batch_abort(dv_batch* batch);                         /* a member thread failed before arriving: every waiting and later dv_batch_arrive returns -1 */
int dv_batch_info(dv_batch* batch, long long* batched_rounds, long long* single_rounds);
/* The FRONT ENDS of the members in shared launches: dv_track_stereo_enqueue for several members at once — FeatureTracker::TrackImage (front_end/background_tracker.cpp:52-158)
 * of S sequences as one launch per stage (pyramid levels, aprons, temporal LK, compaction, Shi-Tomasi tile, corner selection, stereo LK, rows: 10 launches per group and
 * frame instead of 10 per sequence; the reference runs one process per sequence, system/main.cpp:178-330).  member = index into the array dv_batch_create was given.
 * Jobs that cannot share launches (a mode other than DV_MODE_RAW, a mask, BGR frames, installed undistortion maps, an object tracker on the ctx, another image size
 * than the first job's) run through their member's own dv_track_stereo_enqueue inside this call.  Every member is collected with dv_track_stereo_collect as usual;
 * its rows are bit-identical to the unbatched path's. */
typedef struct dv_track_job { int32_t member, mem /* DV_MEM_* [| DV_FMT_BGR] */; const uint8_t* gray0; const uint8_t* gray1; int32_t stride /* bytes, 0 = width */, mode; double t; const uint8_t* mask; } dv_track_job;
int dv_batch_track_enqueue(dv_batch* batch, const dv_track_job* jobs, int n);
int dv_batch_track_info(dv_batch* batch, long long* rounds, long long* members_batched, long long* members_single);
/* measurement: HIP events on the batch stream around the solve / evaluation / reduce launches of one steady-state iteration slot per round; out3 = average ms per launch so far */
int dv_batch_timing(dv_batch* batch, int on, double* out3, long long* rounds, int* windows);

/* Replaces MarginalizationInfo::{preMarginalize,marginalize,getParameterBlocks} as driven by
 * Estimator::SetMarginalizationInfo (estimator/estimator.cpp:403-619).
 *   mode 0 = kMarginOld: P holds the linearisation point (all 11 window states), the residual blocks of the landmarks
 *            anchored in frame 0 (>= 4 observations), imu[0] = the IMUFactor (0,1) (nimu 0 if sum_dt >= 10) and the
 *            current prior; pose 0 / speed-bias 0 / those landmarks are marginalized.
 *   mode 1 = kMarginSecondNew: only the prior is used; pose[kWinSize-1] is marginalized.
 * out_A: n x n (n <= 192), out_b: n.  Block indices of out_prior are already shifted (addr_shift).
 * out_prior->valid = 0 when nothing can be marginalized.  diag4 (may be NULL): c0, smallest pivot of A_mm, failure
 * flag, numerical rank of A'. */
int dv_marginalize(dv_ctx* ctx, const dv_ba_problem* P, int mode, dv_ba_prior* out_prior, double* out_A, double* out_b, double* diag4);

/* operator-level factor evaluation (Evaluate() of the three projection factors / IMUFactor) for parity tests.
 * out: n x 54 doubles = r[2] J_pose_i[2x6] J_pose_j[2x6] J_ex0[2x6] J_ex1[2x6] J_lambda[2] J_td[2] (tangent space) */
int dv_proj_eval(dv_ctx* ctx, const dv_ba_factor* factors, int n, const double* pose_i, const double* pose_j,
                 const double* ex0, const double* ex1, const double* inv_depth, const double* td, double* out);
/* out: 15 whitened residuals followed by the whitened 15 x 30 Jacobian (pose_i 6, sb_i 9, pose_j 6, sb_j 9) */
int dv_imu_eval(dv_ctx* ctx, const dv_ba_imu* imu, double g_norm, const double* pose_i, const double* sb_i,
                const double* pose_j, const double* sb_j, double* out);

/* ---- line and dynamic-object factors: residual + Jacobians, one record per residual block.
 * Jacobians are returned in LOCAL sizes, row-major (pose blocks 6 wide: the reference's 7th column is always zero).
 * Bug-for-bug with the reference where its Jacobian is not the derivative of its residual (see be_obj.hip). ---- */
typedef struct dv_line_factor {          /* lineProjectionFactor(obs_i) + its static sqrt_info (line_projection_factor.h) */
    double obs[4];                       /* the two observed endpoints on the normalised plane: x1 y1 x2 y2 */
    double sqrt_info[4];                 /* 2x2 row-major; NOTE the reference never assigns it (zero) — SURVEY 0.6 */
} dv_line_factor;
/* out[n][34] = r[2] | d r / d pose (2x6) | d r / d ex_pose (2x6) | d r / d orth (2x4)
 * (lineProjectionFactor::Evaluate, estimator/factor/line_projection_factor.cpp:24-159) */
int dv_line_eval(dv_ctx* ctx, const dv_line_factor* factors, int n, const double* pose /* n x 7 */, const double* ex_pose /* n x 7 */,
                 const double* orth /* n x 4 */, double* out);
/* LineOrthParameterization::Plus (estimator/factor/line_parameterization.cpp:9-72): out[n][4] = orth (+) delta */
int dv_line_plus(dv_ctx* ctx, const double* orth, const double* delta, int n, double* out);

typedef struct dv_box_point { double pts_w[3]; double dims[3]; } dv_box_point;      /* BoxEncloseStereoPointFactor(point_w, dims) */
/* out[n][21] = r[3] | d r / d pose_obj (3x6)   (estimator/factor/box_factor.cpp:523-565) */
int dv_box_enclose_eval(dv_ctx* ctx, const dv_box_point* points, int n, const double* pose_obj /* n x 7 */, double* out);
/* out[n][4] = r | d r / d box (1x3)            (BoxDimsFactor, box_factor.cpp:728-743) */
int dv_box_dims_eval(dv_ctx* ctx, const double* dims /* n x 3 */, const double* box /* n x 3 */, int n, double* out);
/* out[n][39] = r[3] | d r / d pose_body (3x6, zero) | d r / d pose_obj (3x6)   (BoxOrientationFactor, box_factor.cpp:752-806) */
int dv_box_orientation_eval(dv_ctx* ctx, const double* R_cioi /* n x 9 */, const double* R_bc /* n x 9 */, const double* pose_body /* n x 7 */,
                            const double* pose_obj /* n x 7 */, int n, double* out);

/* ProjectionInstanceFactor(pts_j, pts_i, velocity_j, velocity_i, td_j, td_i, cur_td) (estimator/factor/project_instance_factor.h:30-60): the
 * reprojection factor of a point ON a moving object — observed in frame j with inverse depth inv_dep_j, carried through the object's poses at j
 * and i into camera i.  The reference ships it switched off (every AddResidualBlock in InstanceManager::AddResidualBlockForJointOpt is commented
 * out, estimator_insts.cpp:1258-1419); it is provided as an operator because BASELINE.json's north_star names it.  112 bytes per record. */
typedef struct dv_inst_proj_factor { double pts_j[3], pts_i[3], vel_j[2], vel_i[2], td_j, td_i, cur_td, pad_; } dv_inst_proj_factor;
/* ProjectionInstanceFactor::Evaluate (project_instance_factor.cpp:27-172); parameter blocks per record: body pose j, body pose i, ex_pose[0],
 * object pose j, object pose i (n x 7 each: p, qx qy qz qw), inv_dep_j (n).
 * out[n][64] = r[2] | J_body_j | J_body_i | J_ex | J_obj_j | J_obj_i (2x6 each, tangent space, row-major) | J_inv_dep (2).
 * Bug-for-bug: J_inv_dep has the reference's + sign and un-compensated pts_j (:166). */
int dv_inst_proj_eval(dv_ctx* ctx, const dv_inst_proj_factor* factors, int n, const double* pose_bj, const double* pose_bi, const double* ex_pose,
                      const double* pose_oj, const double* pose_oi, const double* inv_dep_j, double* out);

/* ---- the per-frame object solve (the numeric part): replaces ceres::Solve inside
 * InstanceManager::Optimization (estimator/estimator_insts.cpp:772-807) for the problem that
 * AddInstanceParameterBlock / AddResidualBlockForInstOpt build (estimator_insts.cpp:989-1245):
 *   variables   per object: para_state[0..kWinSize] (pose, PoseLocalParameterization or — plane_constraint —
 *               PoseConstraintLocalParameterization) and para_box[0] (dims)
 *   residuals   per (object, frame) with a 3-D detection: BoxDimsFactor(detection dims) on para_box with HuberLoss(1.0)
 *               and BoxOrientationFactor(R_cioi, ric[0]) on (body pose, para_state[frame]) without a loss;
 *               per triangulated object point: BoxEncloseStereoPointFactor(p_w, inst.box3d->dims) on
 *               para_state[frame] with HuberLoss(1.0); the dims inside this factor are the values para_box holds
 *               when the solve starts (the reference passes them by value)
 *   options     DENSE_SCHUR + DOGLEG, max_num_iterations = max_iters, wall-clock budget disabled (as dv_ba_solve).
 * The body poses enter BoxOrientationFactor with a zero Jacobian (sic), so they never move; they only count in the
 * parameter-tolerance norm, as they do in ceres.  Blocks without a residual are not variables (ceres removes them).
 * The Hessian of this problem is block diagonal (6x6 per object pose, 3x3 per dims block); the trust region is global. ---- */
typedef struct dv_obj_box {              /* Instance::boxes3d[frame]: one per (object, frame) at most */
    int32_t obj, frame;
    double dims[3];                      /* Box3D::dims of the detection */
    double R_cioi[9];                    /* Box3D::R_cioi(), row-major */
} dv_obj_box;
typedef struct dv_obj_point { int32_t obj, frame; double p_w[3]; } dv_obj_point;      /* FeaturePoint::p_w of a triangulated observation */
typedef struct dv_obj_problem {
    int32_t n_obj, n_boxes, n_points, max_iters;
    int32_t plane_kind;                  /* 0 PoseLocalParameterization, 1 plane constraint with IMU (dz = 0), 2 vision only (dy = 0) */
    int32_t reserved;
    double* state;                       /* n_obj x 11 x 7 [p, qx qy qz qw]   in/out: Instance::para_state */
    double* dims;                        /* n_obj x 3                          in/out: Instance::para_box  */
    const double* body_pose;             /* 11 x 7: body.para_pose */
    double R_bc[9];                      /* body.ric[0], row-major */
    const dv_obj_box* boxes; const dv_obj_point* points;
} dv_obj_problem;
int dv_obj_solve(dv_ctx* ctx, dv_obj_problem* problem, dv_ba_summary* summary);

/* ---- the line-only refinement: replaces ceres::Solve inside Estimator::OptimizationWithOnlyLine
 * (estimator/estimator.cpp:345-395) for the problem AddLineResidualBlock builds (:222-253): one LineOrthParameterization block
 * (body.para_line_features[k], 4 parameters) per triangulated line landmark, one lineProjectionFactor per observation with
 * CauchyLoss(1.0), every pose / extrinsic block constant; DENSE_SCHUR + DOGLEG, max_num_iterations = max_iters.
 * sqrt_info is lineProjectionFactor::sqrt_info (the reference leaves it zero: the solve then returns at once). ---- */
typedef struct dv_line_obs { int32_t line, frame; double obs[4]; } dv_line_obs;      /* LineFeature::line_obs of landmark `line` in window frame `frame` */
typedef struct dv_line_problem {
    int32_t n_lines, n_obs, max_iters, reserved;
    double* orth;                        /* n_lines x 4   in/out: body.para_line_features */
    const double* pose;                  /* 11 x 7: body.para_pose */
    const double* ex_pose;               /* 7: body.para_ex_pose[0] */
    double sqrt_info[4];                 /* 2x2 row-major */
    const dv_line_obs* obs;
} dv_line_problem;
int dv_line_solve(dv_ctx* ctx, dv_line_problem* problem, dv_ba_summary* summary);

/* ---- Estimator (estimator/estimator.h:55-164): IMU buffer + one ProcessMeasurements iteration per call ---- */
typedef struct dv_est_config {          /* para (estimator/vio_parameters.cpp:19-83), cfg flags, extrinsics (utils/parameters.cpp) */
    int32_t use_imu, stereo, plane_constraint, max_iters;      /* imu, num_of_cam==2, plane_constraint, max_num_iterations */
    double keyframe_parallax;           /* pixels; kMinParallax = keyframe_parallax / 460 */
    double init_depth, g_norm, td;      /* INIT_DEPTH, g_norm, td */
    double acc_n, gyr_n, acc_w, gyr_w;
    double ric[2][9], tic[2][3];        /* body_T_cam0 / body_T_cam1: rotation (row-major) and translation */
    /* dynamic mode (cfg::slam == SLAM::kDynamic): the object branch of ProcessImage (estimator.cpp:1562-1622,1653-1676) */
    int32_t dynamic;                    /* 1: dv_est_process_dynamic* run the InstanceManager */
    int32_t use_det3d;                  /* use_det3d: objects are initialised from 3-D detections */
    int32_t instance_init_min_num;      /* instance_init_min_num (viode.yaml:135: 4) */
    int32_t estimate;                   /* bit 0: cfg::is_estimate_ex == 1 (estimate_extrinsic 1: the extrinsics are optimised around the configured ones from the first full window with
                                           |Vs[0]| > 0.2 on — openExEstimation, estimator.cpp:87-95,632), bit 1: cfg::is_estimate_td (estimate_td 1, :98-100).  estimate_extrinsic 2
                                           (no initial guess: CalibrationExRotation, estimator.cpp:1426-1445) is not built.  0 in every shipped YAML.  Refused together with dynamic = 1
                                           (dv_est_create fails): the object branch's factors carry no extrinsic / td Jacobians (the reference's do, estimator.cpp:205) */
    double static_inst_threshold;       /* static_inst_threshold: scene-flow norm above which an object counts as moving (default 10) */
    /* line mode (cfg::use_line): line landmarks in FeatureManager, TriangulateLineMono, OptimizationWithOnlyLine, AddLineResidualBlock (estimator.cpp:224-253,345-395) */
    int32_t use_line, line_min_obs;     /* use_line ; line_min_obs (default 5, vio_parameters.cpp:47-54) */
    double line_sqrt_info[4];           /* lineProjectionFactor::sqrt_info, 2x2 row-major.  The reference never assigns it (zero: SURVEY 0.6) — then the line blocks are
                                           inert and only enter the window solve's parameter-tolerance norm.  Non-zero weights are honoured by the line-only solve;
                                           the window solve refuses them (line e-blocks inside the Schur elimination are not built). */
} dv_est_config;

typedef struct dv_est_state {
    int32_t frame, nonlinear /* solver_flag == kNonLinear */, margin_old /* margin_flag == kMarginOld */, n_landmarks, n_long, iterations;
    double initial_cost, final_cost;
    double window[11][16];              /* body.Ps / Rs (as qx qy qz qw) / Vs / Bas / Bgs per window slot */
} dv_est_state;

int dv_est_create(dv_ctx* ctx, const dv_est_config* cfg);       /* Estimator::Estimator + SetParameter */
/* diagnostics, after dv_debug_set(ctx, "hash_log", 1): per window solve [counter, hash of the states uploaded, hash of the tables uploaded, hash of the states downloaded,
 * hash of the outlier flags consumed, iterations]; two runs of the same sequence must agree row by row */
int dv_est_debug_hash_log(dv_ctx* ctx, unsigned long long* rows6, int cap, int* n_rows);
/* the same on the device side: per fused window solve [counter, hash of the uploaded block as it arrived, of the prior A and b the round reads, of x after the round, of the
 * candidate buffer (gauge-fixed copy), of the control block] */
int dv_ba_debug_dev_log(dv_ctx* ctx, unsigned long long* rows7, int cap, int* n_rows);
/* and per LAUNCH of the round: for every fused solve 16 slots x 5 launch kinds (head evaluation, head reduce, solve, candidate evaluation, candidate reduce) x 16 hashed buffers */
int dv_ba_debug_slot_log(dv_ctx* ctx, unsigned long long* vals, long long cap_vals, long long* n_vals, int* row_len);
int dv_est_reset(dv_ctx* ctx);                                   /* Estimator::ClearState + SetParameter */
int dv_est_input_imu(dv_ctx* ctx, double t, const double* acc, const double* gyr);      /* Estimator::InputIMU */
/* one iteration of Estimator::ProcessMeasurements (estimator.cpp:1786-1863): IMU interval, pre-integration,
 * ProcessImage (keyframe test, triangulation, Optimization on the GPU, marginalization on the GPU, outlier rejection,
 * sliding window).  returns 0 = processed, 1 = IMU data does not cover t yet (feed more, call again), <0 error.
 * The frame's pose is out->window[10] once nonlinear (what SaveBodyTrajectory writes, utils/io/output.cpp:199-227). */
int dv_est_process(dv_ctx* ctx, const dv_feat* feats, int n, double t, dv_est_state* out);
/* the same in two phases, like dv_track_stereo_enqueue/_collect: _begin does the host bookkeeping and ENQUEUES the window
 * solve + marginalization on the ctx's BA stream (returns 0, or 1 = IMU data missing: nothing was started); _end waits,
 * applies the result (outlier rejection, window slide) and fills `out`.  Between the two the caller may enqueue the next
 * frame's tracking and feed IMU samples — thread T2's work overlapping T3's, as in the reference (system/main.cpp:394-404). */
int dv_est_process_begin(dv_ctx* ctx, const dv_feat* feats, int n, double t);
int dv_est_process_end(dv_ctx* ctx, dv_est_state* out);
/* Estimator::IMUAvailable(t + td) (estimator/estimator.h:128-133): 1 = the IMU buffer reaches the frame time, 0 = not yet (the "wait for imu" test of
 * ProcessMeasurements BEFORE it pops the frame from feature_queue, estimator.cpp:1800-1805), <0 error.  Always 1 without an IMU. */
int dv_est_imu_available(dv_ctx* ctx, double t);

/* one entry of FeatureBackground::lines (basic/frontend_feature.h:41-44): the matched line `id` of this frame with the undistorted normalised end points
 * (Line::StartPt / EndPt of FrameLines::un_lines) in the left image and, if matched, in the right one.  The LSD / LBD detector that produces them is upstream. */
typedef struct dv_line_row { uint32_t id; int32_t has_right; double left[4], right[4]; } dv_line_row;      /* x1 y1 x2 y2 */
/* frame.features.lines of the NEXT dv_est_process* call (use_line) */
int dv_est_set_lines(dv_ctx* ctx, const dv_line_row* lines, int n);
/* FeatureManager::line_landmarks after the last frame: per landmark id, start_frame, observation count, is_triangulation, line_plucker (n, v; camera frame of
 * start_frame), ptw1 / ptw2 (the `lines` marker publisher) */
typedef struct dv_line_landmark { int32_t id, start_frame, n_obs, is_triangulation; double plucker[6], ptw1[3], ptw2[3]; } dv_line_landmark;
int dv_est_get_lines(dv_ctx* ctx, dv_line_landmark* out, int cap, int* n_out);
/* FrameLines::UndistortedLineEndPoints (front_end side of TrackImageLine, line_detector): end points (x1 y1 x2 y2 pixels) -> normalised, undistorted */
int dv_undistort_lines(dv_ctx* ctx, const dv_cam* cam, const float* lines_xyxy, int n, double* out_xyxy);

/* ---- the members of Estimator the callbacks and publishers use besides ProcessMeasurements (estimator/estimator.h:55-164) ---- */
/* Estimator::ChangeSensorType (estimator.cpp:697-726; the /vins_imu_switch, /vins_cam_switch callbacks): switching the IMU on restarts the estimator
 * (ClearState + SetParameter), switching it off drops the prior.  use_stereo = 0 is refused (monocular initialisation is out of scope). */
int dv_est_change_sensor_type(dv_ctx* ctx, int use_imu, int use_stereo);
/* latest_time / latest_P / latest_Q (qx qy qz qw) / latest_V: the newest frame's state propagated by every IMU sample fed since — FastPredictIMU inside
 * InputIMU and UpdateLatestStates (estimator.cpp:729-742,1376-1418): what PubLatestOdometry publishes on `imu_propagate`.  Returns 1 while not initialised. */
int dv_est_get_latest(dv_ctx* ctx, double* t, double* P3, double* Q4, double* V3);
/* body.ric / body.tic (row-major rotations, translations of cam0 / cam1) and body.td as the last Double2vector left them: the configured values unless dv_est_config::estimate
 * frees them (what pubOdometry writes out when estimate_extrinsic is on, utils/io/visualization.cpp:94-118).  Any pointer may be NULL. */
int dv_est_get_extrinsics(dv_ctx* ctx, double* ric18, double* tic6, double* td);
/* Health of the marginalization (MarginalizationInfo::marginalize, factor/marginalization_factor.cpp:284-304).  The reference zeroes the eigenvalues
 * <= 1e-8 of A_mm silently; the device skips the LDL^T pivots <= 1e-8 (also negative ones) — the same pseudo-inverse whenever the deficient directions
 * are single columns (a landmark without information), order dependent otherwise.  So the event is counted and readable:
 *   checked   marginalizations whose scalars have come back (the estimator reads them one frame late: they run behind the state download)
 *   clamped   of those, how many skipped at least one pivot
 *   last4     c0, smallest pivot of A_mm, clamp flag (!= 0: skipped), rank of A_mm — of the last one that came back */
int dv_est_get_marg_health(dv_ctx* ctx, long long* checked, long long* clamped, double* last4);

/* ---- the per-frame host loop of the reference's threads T2 (FeatureTrack, system/main.cpp:178-330) + T3 (the estimator thread, :394-404) in C++ inside the library:
 * one call runs n_rounds frames of one or many sequences.  Per sequence the order of pipeline.py: collect tracking(k), IMU up to t_k, dv_est_process_begin(k),
 * enqueue tracking(k+1), IMU up to t_k+1, dv_est_process_end(k) — the front end of frame k+1 overlaps the back end of frame k.  With group_size > 1 the sequences
 * are grouped into dv_batch groups of that size: the begin phases of a group run back to back, ONE dv_batch_enqueue launches the iteration slots of all its window
 * solves, and the host turns to the next group while they run (config 4 of BASELINE.json, "batched").  `threads` host threads each drive their own groups (at most one thread per group; more are not used).
 * Opt-in: dv_runner_set(runner, "teams", 1) before the first run lets threads / groups host threads (a multiple) share the host phases of one group's members between barriers
 * (bit-identical to the single-thread run in tests/test_runner.py; off by default).
 * Round 4 found single members of multi-group runs intermittently leaving their trajectory: a missing workgroup barrier in the accept decision (be_accept_body), exposed when a
 * second group kept the CUs busy; fixed (shared launch: 8 of 30 runs differing before, 0 of 60 after — DESIGN.md 0).  Check every sequence's result when you change this path:
 * bench.py does, and refuses to report a run that fails the check; scripts/dbg/multiseq_first_diff.py compares bit for bit against the single-thread run.
 * The contexts (each with its estimator: dv_est_create) stay the caller's; frames are referenced, not copied (device or host memory: dv_seq_input::mem).
 * dynamic_vins_amd/host/dvins_node.cpp is the ROS-free node built on it (image directory + IMU csv in, `<seq>_<mode>_Odometry.txt` out). */
typedef struct dv_seq_input {
    const uint8_t* const* left; const uint8_t* const* right;      /* [n_frames] gray images of the configured size */
    const double* times; int32_t n_frames, mem /* DV_MEM_HOST / DV_MEM_DEVICE */, stride /* bytes per row, 0 = width */, ba_stride /* 2: only every 2nd tracked frame goes to the back end (system/main.cpp:300-307); 0 / 1: every frame */;
    const double* imu_t; const double* imu_acc; const double* imu_gyr; int32_t n_imu, reserved2;      /* [n_imu], [n_imu][3], [n_imu][3]; n_imu 0 for vision-only */
} dv_seq_input;
/* dynamic mode of a sequence (cfg::slam == kDynamic; the estimator must have been created with dv_est_config::dynamic = 1 and the tracker configured with
 * dv_inst_config): per frame what thread T1 / the perception front end of the reference attaches to the SemanticImage (system/main.cpp:59-171) — the inverse merged
 * instance mask, the detections with their track ids and ROI masks, the 3-D detections (use_det3d) and the disparity map the extra points are sampled from.  The runner
 * then drives the reference's dynamic loop per frame: TrackSemanticImage + InstsTrack (both enqueued together, thread T2), collect, dv_est_process_dynamic_begin_ego,
 * the NEXT frame's tracking, dv_est_process_dynamic_attach (the object branch beside the window solve), dv_est_process_end (estimator.cpp:1562-1676).
 * All arrays have n_frames entries and must outlive the runner; masks of detections are host memory, inv_mask / disp follow their *_mem. */
typedef struct dv_inst_det dv_inst_det; typedef struct dv_box3d dv_box3d;      /* defined with the dynamic-mode entries below */
typedef struct dv_seq_dynamic {
    const uint8_t* const* inv_mask; int32_t mask_mem /* DV_MEM_*: must equal dv_seq_input::mem (frames and mask are handed to dv_track_stereo_enqueue with ONE memory kind; checked) */, mode /* DV_MODE_SEMANTIC (default when 0 is passed is RAW: set it) or DV_MODE_NAIVE */;
    const dv_inst_det* const* dets; const int32_t* n_dets;
    const dv_box3d* const* boxes3d; const int32_t* n_boxes3d;      /* may be NULL (no 3-D detector) */
    const float* const* disp; int32_t disp_mem, disp_stride /* bytes, 0 = 4 * width */; double baseline;      /* disp may be NULL: dv_inst_det::points are handed through */
    const uint32_t* const* right_keys; int32_t right_keys_mem;      /* VIODE: per frame the key image of seg1 (dv_inst_set_right_keys), tightly packed; may be NULL */
    int32_t static_as_background;      /* para::is_static_inst_as_background (vio_parameters.h:86: the reference's default is 1): before tracking frame f the pixels of the instances the
                                          estimator reported static leave the merged mask (dv_est_get_static_instances of the newest back-end frame <= f - 2 -> dv_track_unmask_static;
                                          system/main.cpp:194,217-245).  The reference reads that report across threads without an order; the lag of two frames is this runner's, in every layout */
} dv_seq_dynamic;
/* choice T1: the static-instance report applied to frame f is that of the newest back-end frame <= f - DV_STATIC_REPORT_LAG */
#define DV_STATIC_REPORT_LAG 2
typedef struct dv_runner dv_runner;
/* slam_type naive of a sequence (system/main.cpp:263-265: FeatureTrack -> TrackImageNaive): per frame [n_frames] the inverse merged instance mask (0 = object pixel;
 * VIODE::SetViodeMaskSimple or the detector's SetBackgroundMask) in the frames' memory kind, tracked with `mode` = DV_MODE_NAIVE (GPU tracker's + GPU detector's rules).
 * The back end stays the raw one.  Before the first dv_runner_run; a masked sequence inside a dv_batch group keeps its own tracking launches. */
int dv_runner_set_mask(dv_runner* runner, int seq, const uint8_t* const* inv_mask, int mask_mem, int mode);
int dv_runner_set_dynamic(dv_runner* runner, int seq, const dv_seq_dynamic* dyn);      /* before the first dv_runner_run; the sequence then runs outside dv_batch groups */
/* what the object branch of a dynamic sequence was fed so far: detections, object feature rows, frames with at least one object, fewest detections in a frame */
int dv_runner_dynamic_stats(dv_runner* runner, int seq, long long* detections, long long* object_features, long long* frames_with_objects, int* min_detections);
dv_runner* dv_runner_create(dv_ctx* const* ctxs, const dv_seq_input* seqs, int n_seq, int group_size, int threads);      /* group_size <= 1: no batching */
void dv_runner_destroy(dv_runner* runner);
int dv_runner_run(dv_runner* runner, int n_rounds, double* wall_seconds_or_null);
/* per sequence: the last dv_est_state, the trajectory so far as rows [t, px py pz qx qy qz qw] (one per frame solved in the non-linear phase: what SaveBodyTrajectory
 * writes, utils/io/output.cpp:199-227), the window-solve iterations and frames so far, the number of feature rows of the last tracked frame */
int dv_runner_get(dv_runner* runner, int seq, dv_est_state* last, double* poses8, int cap, int* n_poses, long long* iterations, long long* frames, int* n_rows_last);
/* every frame handed to the back end, initialisation included, as rows [t, px py pz qx qy qz qw, nonlinear]: the lines of `<seq>_<mode>_Odometry.txt` */
int dv_runner_get_frames(dv_runner* runner, int seq, double* rows9, int cap, int* n_rows);
/* diagnostics: per frame handed to the back end [frame index, rows collected from the tracker, FNV-1a hash of those rows' bytes, solver iterations] — two runs of the same
 * sequence must agree entry by entry; where they first differ says whether the tracker's output or only the solve moved (raw mode) */
/* diagnostics: the host's steady clock (seconds) at the end of every frame handed to the back end so far (which = 0), or at which the tracker thread of a dynamic sequence
 * delivered every frame (which = 1): per-frame pacing of a run without cutting it into calls */
int dv_runner_get_frame_clock(dv_runner* runner, int seq, int which, double* seconds, int cap, int* n_out);
int dv_runner_get_row_log(dv_runner* runner, int seq, unsigned long long* rows4, int cap, int* n_rows);
int dv_runner_batch_rounds(dv_runner* runner, long long* batched_rounds, long long* single_rounds);      /* dv_batch_info summed over the groups */
int dv_runner_batch_timing(dv_runner* runner, int on, double* out3, long long* rounds, int* windows);      /* dv_batch_timing of the runner's groups, averaged */
/* switches: "batch_front" (default 1): the members of a dv_batch group are tracked in shared launches (dv_batch_track_enqueue); 0: one set of launches per sequence */
int dv_runner_set(dv_runner* runner, const char* key, int value);
int dv_runner_track_info(dv_runner* runner, long long* rounds, long long* members_batched, long long* members_single);      /* dv_batch_track_info summed over the groups */
const char* dv_runner_error(dv_runner* runner);

/* FeatureManager::point_landmarks for the point-cloud publishers (utils/io/visualization.cpp:214-249): world point = CamToWorld(point * depth, start_frame);
 * in_point_cloud / in_margin_cloud apply PubPointCloud's two selection rules.  key_poses = window[i][0..2] of dv_est_state. */
typedef struct dv_landmark { int32_t id, start_frame, n_obs, solve_flag; double depth, p_w[3]; int32_t in_point_cloud, in_margin_cloud; } dv_landmark;
int dv_est_get_landmarks(dv_ctx* ctx, dv_landmark* out, int cap, int* n_out);

/* ---- dynamic mode: the object (instance) half of the back end.  Inputs are FrontendFeature::instances (basic/frontend_feature.h:58-75), i.e.
 * what InstsFeatManager::Output() hands over (front_end/dynamic_tracker.cpp:521-577): per object its tracked features, the associated 3-D
 * detection (if any) and the "extra" 3-D points sampled from the disparity map (camera frame).  The association of detections to tracks
 * (DeepSORT / VIODE keys), the detector networks and the PCL clustering of the extra points are upstream of the path. ---- */
typedef struct dv_box3d {               /* Box3D (basic/box3d.h:40-106): the fields the path reads */
    int32_t class_id, pad_; double score;
    double center[3];                   /* center_pt, camera frame */
    double dims[3]; double yaw;         /* R_cioi() is built from yaw (box3d.h:79-83) */
    float rect_min[2], rect_max[2];     /* box2d of the projected corners (BoxAssociate2Dto3D, dynamic_tracker.cpp:61-152) */
} dv_box3d;
typedef struct dv_inst_obs {            /* one FeatureInstance */
    uint32_t id; int32_t has_box3d;     /* map key (Box2D::track_id); box3d valid */
    int32_t first_feat, n_feats;        /* rows of the instance feature array: dv_feat with left = (x_n, y_n, 1, u, v, vx, vy), right likewise if has_right */
    int32_t first_point, n_points;      /* FeatureInstance::points: (x, y, z) triples in the camera frame */
    float rect[4];                      /* Box2D::rect x, y, w, h */
    dv_box3d box3d;
} dv_inst_obs;
typedef struct dv_inst_state {          /* Instance (estimator/instance.h) as the publishers read it + InstEstimatedInfo (basic/inst_estimated_info.h) */
    uint32_t id; int32_t is_initial, is_tracking, is_curr_visible, is_static, is_init_velocity, age, lost_number, static_frame,
             n_landmarks, n_valid, triangle_num;
    double dims[3], vel_v[3], vel_a[3];
    double window[11][7];               /* state[i]: P, then R as qx qy qz qw */
    double time[11];
} dv_inst_state;
/* ---- dynamic mode, front end: InstsFeatManager (front_end/dynamic_tracker.h:42-101).  One detection = one Box2D of SemanticImage::boxes2d AFTER the
 * multi-object tracker assigned its track id (DeepSORT / VIODE keys: upstream).  mask = InstRoi::mask_cv, roi_gray is cropped from the frame by the
 * library.  points (optional) = InstFeat::extra_points3d of this frame computed by the caller, handed through to the output — or give the library the frame's
 * disparity map (dv_inst_set_disparity) and it runs DetectExtraPoints + ProcessExtraPoints itself. */
typedef struct dv_inst_det {
    uint32_t track_id; int32_t class_id;
    int32_t x, y, w, h;                 /* Box2D::rect, inside the image */
    const uint8_t* mask;                /* h rows of w bytes, > 0 = object (host memory) */
    const double* points; int32_t n_points, pad_;      /* ignored for a frame whose disparity map was handed over (dv_inst_set_disparity) */
} dv_inst_det;
/* fe_para::kMaxDynamicCnt / kMinDynamicDist (front_end/front_end_parameters.cpp), cfg::use_det3d; call once before the first frame */
int dv_inst_config(dv_ctx* ctx, int max_dynamic_cnt, int min_dynamic_dist, int use_det3d);
int dv_inst_reset(dv_ctx* ctx);
/* FeatureTrack's object branch for one frame (system/main.cpp:198-250): AddInstancesByTracking + InstsFeatManager::InstsTrack.  Works on the pyramids
 * of the frame last passed to dv_track_stereo_enqueue (call it right after, same frame; `t` = that frame's time).  boxes3d = SemanticImage::boxes3d
 * (only read when use_det3d). */
int dv_inst_track_enqueue(dv_ctx* ctx, double t, const dv_inst_det* dets, int n_dets, const dv_box3d* boxes3d, int n_boxes3d);
/* SemanticImage::disp (CV_32F disparity of the left image, basic/semantic_image.h:30-65) of the frame about to be handed to dv_inst_track_enqueue.  With it the library
 * runs the reference's extra-point pipeline for every visible object ON THE DEVICE, one launch for all objects on a side stream (the reference: a second thread):
 * InstFeat::DetectExtraPoints (front_end/instance_feature.cpp:413-461: step = max(sqrt(0.8 rows cols / 1000), 2), mask > 0, disparity > 0 and not NaN,
 * depth = fx0 baseline / disparity in (0.1, 100], x = (c - cx0) depth / fx0 ..., float arithmetic, row-major scan order) and the point-cloud half of
 * InstsFeatManager::ProcessExtraPoints (front_end/dynamic_tracker.cpp:268-338: pcl::RadiusOutlierRemoval(0.5, 10), pcl::EuclideanClusterExtraction(1.0, 10, 25000), first
 * cluster).  dv_inst_det::points is then ignored; dv_inst_track_collect returns the computed points.  disp: rows of `stride_bytes` bytes (0 = 4 * width), config size,
 * host or device memory (a device map must stay valid until dv_inst_track_collect); baseline = cam_s.baseline (utils/camera_model.h:38); fx0, fy0, cx0, cy0 = cam0 of the
 * config as float.  disp == NULL: back to the pass-through form.  Applies to the NEXT dv_inst_track_enqueue only. */
int dv_inst_set_disparity(dv_ctx* ctx, const float* disp, int stride_bytes, int mem, double baseline);
/* cfg::dataset == kViode: the keys (VIODE::PixelToKey; dv_viode_mask's key_image) of SemanticImage::seg1 — the RIGHT camera's segmentation image — of the frame the NEXT
 * dv_inst_track_enqueue processes.  InstFeat::TrackRightByPad then keeps a right-image point only where that image carries the object's key (dv_inst_det::track_id):
 * status[i] && VIODE::PixelToKey(right_points[i], img.seg1) != id -> 0 (front_end/instance_feature.cpp:263-268).  w x h uint32, stride in bytes (0 = 4 w), mem DV_MEM_*;
 * host memory must stay valid until the frame is collected.  NULL: no test (KITTI / custom data sets).  Belongs to one frame, like the disparity map. */
int dv_inst_set_right_keys(dv_ctx* ctx, const uint32_t* key_image, int stride_bytes, int mem);
/* operator form of the same pipeline for one object (parity tests): mask = h rows of w bytes (host), (x, y) = Box2D::rect.tl().  stage 0: the whole pipeline (the segmented
 * cloud); stage 1: InstFeat::DetectExtraPoints alone (the sampled points before any filtering).  out_xyz: cap_out triples at most, the float results widened to double. */
int dv_extra_points(dv_ctx* ctx, const uint8_t* mask, int x, int y, int w, int h, const float* disp, int stride_bytes, int mem, double baseline, int stage,
                    double* out_xyz, int cap_out, int* n_out);
/* InstsFeatManager::Output() (front_end/dynamic_tracker.cpp:521-577): waits for the frame; insts / feats / points are laid out as dv_est_process_dynamic takes them */
int dv_inst_track_collect(dv_ctx* ctx, dv_inst_obs* insts, int cap_insts, int* n_insts, dv_feat* feats, int cap_feats, int* n_feats,
                          double* points, int cap_points, int* n_points);

/* ProcessMeasurements iteration in dynamic mode: dv_est_process with frame.instances.  insts may be NULL / n_insts 0 (no object in view). */
int dv_est_process_dynamic(dv_ctx* ctx, const dv_feat* feats, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats,
                           const double* points, dv_est_state* out);
/* two-phase form: _begin enqueues the window solve first and runs the object branch (bookkeeping on the host, InstanceManager::Optimization
 * as dv_obj_solve on a third stream) while it is in flight; dv_est_process_end collects both. */
int dv_est_process_dynamic_begin(dv_ctx* ctx, const dv_feat* feats, int n, double t, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats,
                                 const double* points);
/* three-phase form: _begin_ego enqueues the window solve with the background features alone; _attach hands over the frame's instances and runs the object branch
 * while the window solve is in flight (a frame that is never attached is processed as a frame without objects); dv_est_process_end collects both.  Same results
 * as the two-phase form: what the caller does between the calls (enqueueing the next frame's tracking, dv_inst_track_collect) leaves the front of the window solve. */
int dv_est_process_dynamic_begin_ego(dv_ctx* ctx, const dv_feat* feats, int n, double t);
int dv_est_process_dynamic_attach(dv_ctx* ctx, const dv_inst_obs* insts, int n_insts, const dv_feat* inst_feats, const double* points);
/* Estimator::im.instances after the last processed frame (ascending id); *n_out = number written (<= cap); summary4 (may be NULL): iterations,
 * termination, initial and final cost of the last object solve */
int dv_est_get_instances(dv_ctx* ctx, dv_inst_state* out, int cap, int* n_out, double* summary4);
/* InstanceManager::GetOutputInstInfo as far as the front end reads it (estimator_insts.cpp:967-990; system/main.cpp:194,217-245): the ids of the initialised, tracked
 * instances that were is_static when the LAST dynamic frame took its snapshot — right behind PushBack (estimator.cpp:1579-1586), so with the flags the frame before
 * left.  Ascending.  The tracker side of the feedback is dv_track_unmask_static. */
int dv_est_get_static_instances(dv_ctx* ctx, uint32_t* ids, int cap, int* n_out);

/* ---- measurement hooks (used by bench.py; HIP-event timing on the ctx's own stream) ---- */
/* names: "pyr","lk_temporal","compact","gftt_eig","gftt_select","lk_stereo","frame" */
/* on: 0 off, 1 per-stage events, 2 additionally one event pair around every back-end kernel launch
 * ("k_be_eval_full","k_be_reduce","k_be_solve","k_be_eval_cost","k_be_accept","k_be_marg"), -1 host wall-clock scopes only
 * ("h_*": no events, no extra synchronisation — the pipeline keeps its overlap) */
int dv_timing_enable(dv_ctx* ctx, int on);
int dv_timing_reset(dv_ctx* ctx);
int dv_timing_get(dv_ctx* ctx, const char* name, double* total_ms, long long* count);

/* debug-only switch for the test-suite (never read from the environment).  key "short_first_pass": the window solve enqueues
 * max_iters - 2 slots first, so the spare-slot continuation (rare in production: only after a failed linear solve) runs every frame. */
int dv_debug_set(dv_ctx* ctx, const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif
