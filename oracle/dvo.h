/*
 * dvo.h — C API of the CPU ORACLE (test infrastructure, NOT the product).
 *
 * The oracle is a dependency-free C++17 restatement of the reference's hot path
 * (dynamic_vins front-end tracking + sliding-window BA).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (dynamic_vins_amd/, libdvins_hip.so) never links, imports or calls anything here.
 *
 * PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be
 * built in this image (needs ROS/OpenCV-CUDA/Ceres/Eigen).  The third-party pieces
 * (OpenCV 3.4.16 calcOpticalFlowPyrLK / goodFeaturesToTrack / circle, Ceres 1.14
 * DENSE_SCHUR+DOGLEG) are restated from their published algorithms;
 * first-party pieces cite the reference file:line they follow.
 */
#ifndef DVO_H
#define DVO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* CPU-baseline timing only: threads for the per-point LK loop (OpenCV: parallel_for_) and, if > 1, the reference's 4-thread build of the marginalization
 * system (marginalization_factor.h:26).  Tracking results do not depend on it; the marginalization sum order does (last bits). Default 1. */
void dvo_set_threads(int n);
/* SENSITIVITY variants of the restatement (tests/test_oracle_variants.py): the third-party arithmetic this oracle restates from memory ("unverified")
 * is platform dependent or uncertain in a few places; the canonical choices are D1, D2 and A.3's radius rule.  A variant swaps ONE of them for the other
 * plausible reading so that a test can BOUND what the choice is worth (features selected, trajectory) instead of asserting it.  Default 0 everywhere = canonical.
 *   "lk_sums":   0 exact int64 window sums (D1) | 1 float accumulation in OpenCV's scalar (raster) order | 2 four float partial sums by x mod 4, combined (a 4-wide SIMD order)
 *   "box_sums":  0 3x3 covariance box sums in double, fixed order (D2) | 1 in float
 *   "f4_cpu_rule": 0 the reference's GPU tracker (cv::cuda::SparsePyrLKOpticalFlow restated, lk_cuda.cpp) where the reference uses it | 1 the CPU LK arithmetic with FeatureTrackByLKGpu's 1.0 px threshold
 *   "radius":    0 Ceres' dogleg radius rule (quality < 0.25: x 0.5; > 0.75: max(radius, 3 |step|)) | 1 the Levenberg-style reading (> 0.75: radius x 3)
 *   "f5_cpu_rule": 0 the reference's GPU corner detector (cv::cuda::GoodFeaturesToTrackDetector restated, gftt_cuda.cpp) in TrackImageNaive | 1 cv::goodFeaturesToTrack's rule there (rounds 1-5)
 *   "gftt_cuda_fma": 0 the GPU detector's float chains contracted to fused multiply-adds (nvcc's default) | 1 no contraction
 *   "gftt_cuda_tie": 0 corners of equal response in address-descending order | 1 ascending (the library's order is the kernel schedule)
 *   "obj_point_order": 0 the object solve's point blocks in the reference's order | 1 reversed (summation order only: what rounding noise is worth to the object states)
 *   "obj_perturb": 0 < n < 1000 moves the world points the object solve reads by n x 1e-7 m (per-frame input noise); 1000 + n moves the body positions the WHOLE object branch sees by
 *                  n x 1e-7 m, keyed to the frame's time stamp (what the ego-state difference between two correct window solves is worth to the objects: tests/tools/obj_sensitivity.py) */
void dvo_set_variant(const char* key, int value);
int dvo_get_variant(const char* key);
int dvo_get_threads(void);

/* ---------------- front end ---------------- */

typedef struct dvo_cam {   /* camodocal PinholeCamera parameters (PinholeCamera.cc:292-295) */
    double fx, fy, cx, cy, k1, k2, p1, p2;
} dvo_cam;

typedef struct dvo_fe_config {  /* fe_para (front_end_parameters.cpp:18-40) + cfg::is_stereo */
    int width, height;
    int max_cnt;       /* fe_para::kMaxCnt  */
    int min_dist;      /* fe_para::kMinDist */
    int flow_back;     /* fe_para::is_flow_back */
    int stereo;        /* cfg::is_stereo */
    dvo_cam cam0, cam1;
} dvo_fe_config;

/* one row per tracked feature, left observation always present */
typedef struct dvo_feat {
    uint32_t id;
    int32_t track_cnt;
    int32_t has_right;
    int32_t pad_;
    double left[7];   /* x_n, y_n, 1, u, v, vx, vy  (background_tracker.cpp:347-355) */
    double right[7];
} dvo_feat;

/* cv::pyrDown 8U, 5x5 [1 4 6 4 1]^2 /256, BORDER_REFLECT_101; dst is ((w+1)/2)x((h+1)/2) */
void dvo_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst);
/* VIODE label image -> masks / key image / per-key bounding boxes (utils/dataset/viode_utils.cpp:21-170) */
void dvo_viode_mask(const uint8_t* seg_bgr, int w, int h, int stride, const uint32_t* dyn_keys, int nkeys, uint8_t* merge, uint8_t* inv, uint32_t* key_img, int32_t* boxes);
/* cv::cvtColor(BGR2GRAY) 8U (SemanticImage::SetGrayImage, basic/semantic_image.cpp:95-99); gray tightly packed */
void dvo_bgr2gray(const uint8_t* bgr, int w, int h, int stride, uint8_t* gray);
/* cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with fixed-point maps (CV_16SC2 + CV_16UC1), 8-bit, cn = 1 or 3; dst is dense (w * cn per row) */
void dvo_remap(const uint8_t* src, int w, int h, int stride, int cn, const int16_t* map1_xy, const uint16_t* map2, uint8_t* dst);
/* cv::initUndistortRectifyMap(K, D, I, newK, size, CV_16SC2) for a pinhole radtan camera */
void dvo_init_undistort_map(const dvo_cam* cam, const double* newK4, int w, int h, int16_t* map1_xy, uint16_t* map2);

/* calcScharrDeriv (OpenCV lkpyramid.cpp): out is int16 interleaved (Ix,Iy), w*h*2 */
void dvo_scharr(const uint8_t* src, int w, int h, int16_t* out);

/* cv::calcOpticalFlowPyrLK(img_a,img_b,pts_a,pts_b,status,err,Size(21,21),max_level,
 * TermCriteria(COUNT+EPS,iters,eps), use_initial?OPTFLOW_USE_INITIAL_FLOW:0, 1e-4).
 * pts are interleaved (x,y) float; pts_b is in/out when use_initial. */
void dvo_lk(const uint8_t* img_a, const uint8_t* img_b, int w, int h,
            const float* pts_a, int n, int max_level, int iters, double eps,
            int use_initial, float* pts_b, uint8_t* status);

/* FeatureTrackByLK (front_end/feature_utils.cpp:35-69); dist_thresh 0.5 there */
void dvo_track_by_lk(const uint8_t* img1, const uint8_t* img2, int w, int h,
                     const float* pts1, int n, int flow_back, float dist_thresh,
                     float* pts2, uint8_t* status);

/* the reference's GPU tracker (lk_cuda.cpp): cuda::pyrDown on 8-bit images (round half to even), one bilinear texture fetch (8-bit fractions, normalised read, clamp),
 * cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), max_level, iters, use_initial)->calc, and FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163: forward + backward,
 * distance <= 1.0, InBorder) */
void dvo_pyr_down_cuda(const uint8_t* src, int w, int h, uint8_t* dst);
float dvo_tex_read(const uint8_t* img, int w, int h, float x, float y);
void dvo_lk_cuda(const uint8_t* img_a, const uint8_t* img_b, int w, int h, const float* pts_a, int n, int max_level, int iters, int use_initial, float* pts_b, uint8_t* status);
void dvo_track_by_lk_gpu(const uint8_t* img1, const uint8_t* img2, int w, int h, const float* pts1, int n, int flow_back, float* pts2, uint8_t* status);

/* cv::cornerMinEigenVal(img, eig, 3, 3) */
void dvo_min_eigen(const uint8_t* img, int w, int h, float* eig);

/* cv::goodFeaturesToTrack(img, out, max_n, quality, min_dist, mask) ; mask may be NULL */
void dvo_gftt(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n,
              double quality, double min_dist, float* out_xy, int* n_out);

/* cv::cuda::createGoodFeaturesToTrackDetector(CV_8UC1, max_n, quality, min_dist)->detect(img, out, mask) (DetectShiTomasiCornersGpu, feature_utils.cpp:339-348)
 * and its response map cuda::createMinEigenValCorner(CV_8UC1, 3, 3)->compute (gftt_cuda.cpp) */
void dvo_gftt_cuda(const uint8_t* img, const uint8_t* mask, int w, int h, int max_n, double quality, double min_dist, float* out_xy, int* n_out);
void dvo_min_eigen_cuda(const uint8_t* img, int w, int h, float* eig);

/* cv::circle(mask, Point(cvRound(x),cvRound(y)), radius, 0, -1) for each point */
void dvo_circle_mask(uint8_t* mask, int w, int h, const float* pts_xy, int n, int radius);

/* cv::erode with k x k rect, anchor centre, border +inf (feature_utils.h:142-146) */
void dvo_erode(const uint8_t* src, int w, int h, int k, uint8_t* dst);

/* PinholeCamera::liftProjective (PinholeCamera.cc:450-508) -> (x/z, y/z) as float */
void dvo_lift_projective(const dvo_cam* cam, const float* pts_xy, int n, float* out_xy);

/* FeatureTracker (front_end/background_tracker.cpp) */
typedef struct dvo_tracker dvo_tracker;
dvo_tracker* dvo_tracker_create(const dvo_fe_config* cfg);
void dvo_tracker_destroy(dvo_tracker*);
/* TrackImage (raw mode, :52-158). mask==NULL. out must hold max_cnt rows. returns n */
int dvo_tracker_track_image(dvo_tracker*, const uint8_t* gray0, const uint8_t* gray1,
                            double time, dvo_feat* out);
/* TrackImageNaive-style (:400-516): mask = inv_merge_mask (0 = object), eroded by caller;
 * uses the A.1 LK (canonical) with fwd/bwd threshold 1.0 (feature_utils.cpp:126) and
 * DetectNewFeature's "<10 -> skip" refill rule (instance_feature.cpp:353-356). */
int dvo_tracker_track_image_naive(dvo_tracker*, const uint8_t* gray0, const uint8_t* gray1,
                                  const uint8_t* mask, double time, dvo_feat* out);
/* mode 0 TrackImage, 1 TrackImageNaive, 2 TrackSemanticImage; erode_k > 0: mask eroded by a k x k rectangle first */
int dvo_tracker_track_image_mode(dvo_tracker*, const uint8_t* gray0, const uint8_t* gray1, const uint8_t* inv_mask_or_null, int mode, int erode_k, double time, dvo_feat* out);
/* one visible object instance through one frame of InstsFeatManager::InstsTrack (front_end/dynamic_tracker.cpp:348-470); see front_oracle.cpp */
int dvo_inst_track(const uint8_t* prev_roi, int pw, int ph, const uint8_t* cur_roi, int cw, int ch, const uint8_t* cur_mask, int box_x, int box_y,
                   const uint8_t* gray0, const uint8_t* gray1, int W, int H, const dvo_cam* cam0, const dvo_cam* cam1,
                   int n_last, const float* last_pts, const uint32_t* ids, const int32_t* track_cnt,
                   int max_cnt, int min_dist, int flow_back, uint32_t* global_id,
                   int* n_cur, float* cur_pts, uint32_t* cur_ids, int32_t* cur_cnt, float* cur_un,
                   int* n_right, float* right_pts, uint32_t* right_ids, float* right_un);

/* ---------------- back end ---------------- */

/* projection factors: kind 0 = ProjectionTwoFrameOneCamFactor (blocks pose_i,pose_j,ex0,lambda,td),
 * 1 = ProjectionTwoFrameTwoCamFactor (pose_i,pose_j,ex0,ex1,lambda,td), 2 = ProjectionOneFrameTwoCamFactor
 * (ex0,ex1,lambda,td).  obs12 = pts_i(3) pts_j(3) vel_i(2) vel_j(2) td_i td_j.  J[k] row-major
 * 2 x block_size (may be NULL entries / NULL). */
void dvo_proj_eval(int kind, const double* obs12, const double* const* par, double* res2, double** J);

/* IntegrationBase (estimator/imu/integration_base.h) */
typedef struct dvo_preint dvo_preint;
dvo_preint* dvo_preint_create(const double* acc0, const double* gyr0, const double* ba, const double* bg,
                              const double* noise4 /* acc_n gyr_n acc_w gyr_w */);
void dvo_preint_destroy(dvo_preint*);
void dvo_preint_push(dvo_preint*, double dt, const double* acc, const double* gyr);
void dvo_preint_repropagate(dvo_preint*, const double* ba, const double* bg);
void dvo_preint_get(const dvo_preint*, double* sum_dt, double* dp, double* dq_xyzw, double* dv, double* jac225, double* cov225);
/* overwrite the integration results (tests: evaluate an IMUFactor for given deltas / Jacobian / covariance) */
void dvo_preint_set(dvo_preint*, double sum_dt, const double* dp, const double* dq_xyzw, const double* dv, const double* jac225, const double* cov225);
/* IMUFactor::Evaluate; par = pose_i(7) sb_i(9) pose_j(7) sb_j(9); J = 15x7,15x9,15x7,15x9 row-major */
void dvo_imu_eval(const dvo_preint*, double g_norm, const double* const* par, double* res15, double** J);

/* line and dynamic-object factors (obj_factors.cpp); Jacobians in the reference's global block sizes, row-major, any J[k] may be NULL */
void dvo_line_eval(const double* obs4, const double* sqrt_info4, const double* const* par /* pose7, ex7, orth4 */, double* res2, double** J /* 2x7, 2x7, 2x4 */);
void dvo_line_plus(const double* orth4, const double* delta4, double* out4);      /* LineOrthParameterization::Plus */
/* line geometry + two-view line triangulation (line_detector/line_geometry.cpp:75-296, estimator/vio_util.cpp:447-561) */
void dvo_plk_to_orth(const double* plk6, double* orth4);
void dvo_orth_to_plk(const double* orth4, double* plk6);
int dvo_line_trimming(const double* plk6, const double* obs4, double* p1, double* p2);
int dvo_triangulate_line(const double* obs, int nobs, int start_frame, const double* Rs, const double* Ps, const double* ric9, const double* tic3,
                         double* plk6, double* ptw1, double* ptw2);
/* ProjectionInstanceFactor::Evaluate (estimator/factor/project_instance_factor.cpp:27-172; dead code in the reference, named by north_star).
 * obs12 = pts_j(3) pts_i(3) vel_j(2) vel_i(2) td_j td_i; par = pose_bj, pose_bi, ex_bc, pose_oj, pose_oi (7 each), inv_dep_j; J = 5 x (2x7) + 2x1 */
void dvo_inst_proj_eval(const double* obs12, double cur_td, const double* const* par, double* res2, double** J);
void dvo_box_enclose_eval(const double* pts_w3, const double* dims3, const double* const* par /* pose_obj7 */, double* res3, double** J /* 3x7 */);
void dvo_box_dims_eval(const double* dims3, const double* const* par /* box3 */, double* res1, double** J /* 1x3 */);
void dvo_box_orientation_eval(const double* R_cioi9, const double* R_bc9, const double* const* par /* pose_body7, pose_obj7 */, double* res3, double** J /* 3x7, 3x7 */);

/* the per-frame object solve (InstanceManager::Optimization, estimator/estimator_insts.cpp:772-807); layout identical to
 * include/dvins.h dv_obj_* (restated here: the oracle shares no headers with the product) */
typedef struct dvo_obj_box { int32_t obj, frame; double dims[3]; double R_cioi[9]; } dvo_obj_box;
typedef struct dvo_obj_point { int32_t obj, frame; double p_w[3]; } dvo_obj_point;
typedef struct dvo_obj_problem {
    int32_t n_obj, n_boxes, n_points, max_iters, plane_kind, reserved;
    double* state; double* dims; const double* body_pose; double R_bc[9];
    const dvo_obj_box* boxes; const dvo_obj_point* points;
} dvo_obj_problem;
struct dvo_ba_summary;
int dvo_obj_solve(dvo_obj_problem* problem, struct dvo_ba_summary* summary);
/* the line-only refinement (Estimator::OptimizationWithOnlyLine, estimator/estimator.cpp:345-395); layout identical to include/dvins.h dv_line_* */
typedef struct dvo_line_obs { int32_t line, frame; double obs[4]; } dvo_line_obs;
typedef struct dvo_line_problem { int32_t n_lines, n_obs, max_iters, reserved; double* orth; const double* pose; const double* ex_pose; double sqrt_info[4]; const dvo_line_obs* obs; } dvo_line_problem;
int dvo_line_solve(dvo_line_problem* problem, struct dvo_ba_summary* summary);

/* flat window problem, identical layout to include/dvins.h dv_ba_* (restated here: the oracle shares no headers
 * with the product) */
typedef struct dvo_ba_factor { double pix, piy, pjx, pjy, vix, viy, vjx, vjy, td_i, td_j; int32_t kind, lm, fi, fj; double pad_[2]; } dvo_ba_factor;
typedef struct dvo_ba_lm { int32_t first, count, anchor, mask; } dvo_ba_lm;
typedef struct dvo_ba_imu { double sum_dt, dp[3], dq[4], dv[3], lin_ba[3], lin_bg[3]; double jacobian[225], covariance[225]; int32_t fi, fj, pad0, pad1; } dvo_ba_imu;
typedef struct dvo_ba_prior_block { int32_t type, idx, off, size_local; } dvo_ba_prior_block;
typedef struct dvo_ba_prior { int32_t valid, n, nblocks, pad; double c0; dvo_ba_prior_block blocks[16]; double x0[16][9]; } dvo_ba_prior;
typedef struct dvo_ba_problem {
    int32_t nframes, nlm, nfac, nimu, use_imu, plane_kind, max_iters, free_blocks;      /* free_blocks: bit 0 para_ex_pose not SetParameterBlockConstant, bit 1 para_td (estimator.cpp:88-100) */
    double g_norm;
    double *pose, *speed_bias, *ex_pose, *td, *inv_depth;
    const dvo_ba_factor* factors; const dvo_ba_lm* landmarks; const dvo_ba_imu* imu;
    const dvo_ba_prior* prior; const double* prior_A; const double* prior_b;
    double x_norm2_extra;            /* squared norm of parameter blocks that are in ceres' x but carry no live residual (the line blocks); layout shared with dv_ba_problem */
} dvo_ba_problem;
typedef struct dvo_ba_summary { int32_t iterations, successful, termination, slots; double initial_cost, final_cost; } dvo_ba_summary;
/* ceres::Solve restatement on a standalone window (Estimator::Optimization's problem, estimator.cpp:261-326) */
int dvo_ba_solve(dvo_ba_problem* problem, dvo_ba_summary* summary);
double dvo_prior_c0(const double* A, const double* b, int n);
/* SetMarginalizationInfo on a flat window (estimator.cpp:403-619): mode 0 kMarginOld, 1 kMarginSecondNew; output in
 * information form A = J0^T J0, b = J0^T r0, c0 = r0^T r0, blocks in the oracle's insertion order (M1) */
int dvo_marginalize(const dvo_ba_problem* P, int mode, dvo_ba_prior* out, double* out_A, double* out_b);

typedef struct dvo_be_config {       /* para (estimator/vio_parameters.cpp:19-83) + cfg flags + extrinsics */
    int use_imu, stereo, plane_constraint, max_iters;
    double keyframe_parallax;        /* pixels; min_parallax = keyframe_parallax / 460 */
    double init_depth, g_norm, td;
    double acc_n, gyr_n, acc_w, gyr_w;
    double ric[2][9], tic[2][3];     /* body_T_cam0 / body_T_cam1 rotation (row-major) and translation */
    int dynamic, use_det3d, instance_init_min_num, estimate;      /* cfg::slam == kDynamic; use_det3d; para::kInstanceInitMinNum; estimate: bit 0 cfg::is_estimate_ex, bit 1 cfg::is_estimate_td */
    double static_inst_threshold;    /* para::kStaticInstThreshold */
    int use_line, line_min_obs;      /* cfg::use_line, para::kLineMinObs */
    double line_sqrt_info[4];        /* lineProjectionFactor::sqrt_info, row-major 2x2 (the reference never assigns it: zero) */
} dvo_be_config;
typedef struct dvo_line_row { uint32_t id; int32_t has_right; double left[4], right[4]; } dvo_line_row;      /* FeatureBackground::lines entry */
typedef struct dvo_line_landmark { int32_t id, start_frame, n_obs, is_triangulation; double plucker[6], ptw1[3], ptw2[3]; } dvo_line_landmark;

typedef struct dvo_be_state {
    int frame, nonlinear, margin_old, n_landmarks, n_long, iterations;
    double initial_cost, final_cost;
    double window[11][16];           /* per window slot: P(3) Q(xyzw) V(3) Ba(3) Bg(3) */
} dvo_be_state;

/* Estimator (estimator/estimator.cpp): InputIMU + one ProcessMeasurements iteration per call.
 * returns 0 = processed, 1 = IMU data does not yet cover t (call again after more InputIMU). */
typedef struct dvo_estimator dvo_estimator;
dvo_estimator* dvo_estimator_create(const dvo_be_config*);
void dvo_estimator_destroy(dvo_estimator*);
void dvo_estimator_input_imu(dvo_estimator*, double t, const double* acc, const double* gyr);
int dvo_estimator_process(dvo_estimator*, const dvo_feat* feats, int n, double t, dvo_be_state* out);

/* dynamic mode: FrontendFeature::instances as flat arrays (layout identical to include/dvins.h dv_box3d / dv_inst_obs / dv_inst_state, restated here) */
typedef struct dvo_box3d { int32_t class_id, pad_; double score; double center[3]; double dims[3]; double yaw; float rect_min[2], rect_max[2]; } dvo_box3d;
typedef struct dvo_inst_obs { uint32_t id; int32_t has_box3d; int32_t first_feat, n_feats; int32_t first_point, n_points; float rect[4]; dvo_box3d box3d; } dvo_inst_obs;
typedef struct dvo_inst_state {
    uint32_t id; int32_t is_initial, is_tracking, is_curr_visible, is_static, is_init_velocity, age, lost_number, static_frame, n_landmarks, n_valid, triangle_num;
    double dims[3], vel_v[3], vel_a[3]; double window[11][7]; double time[11];
} dvo_inst_state;
/* InstsFeatManager (front_end/dynamic_tracker.{h,cpp}); shares the id counter and the configuration of a dvo_tracker; see front_oracle.cpp */
typedef struct dvo_inst_det { uint32_t track_id; int32_t class_id; int32_t x, y, w, h; const uint8_t* mask; const double* points; int32_t n_points, pad_; } dvo_inst_det;
typedef struct dvo_insts dvo_insts;
dvo_insts* dvo_insts_create(dvo_tracker* background, int max_dynamic_cnt, int min_dynamic_dist, int use_det3d);
void dvo_insts_destroy(dvo_insts*);
int dvo_insts_track(dvo_insts*, const uint8_t* gray0, const uint8_t* gray1, double time, const dvo_inst_det* dets, int n_dets, const dvo_box3d* boxes3d, int n_boxes3d);
/* SemanticImage::disp (CV_32F, width x height of the tracker) of the NEXT dvo_insts_track call: the extra points of every visible object are then computed from it
 * (InstFeat::DetectExtraPoints + the PCL half of ProcessExtraPoints, extra_points.cpp) and dvo_inst_det::points is ignored.  baseline = cam_s.baseline.  The map is copied. */
void dvo_insts_set_disparity(dvo_insts*, const float* disp, float baseline);
/* VIODE: the keys (VIODE::PixelToKey, dvo_viode_mask's key image) of SemanticImage::seg1 of the NEXT dvo_insts_track call: TrackRightByPad drops a right-image point whose
 * pixel does not carry the object's key (front_end/instance_feature.cpp:263-268).  NULL: no test (the other datasets).  The image is copied. */
void dvo_insts_set_right_keys(dvo_insts*, const uint32_t* key_img);
/* InstFeat::DetectExtraPoints (front_end/instance_feature.cpp:413-461) / the point-cloud half of InstsFeatManager::ProcessExtraPoints (front_end/dynamic_tracker.cpp:268-338) */
int dvo_detect_extra_points(const uint8_t* mask, int cols, int rows, int box_x, int box_y, const float* disp, int disp_w, int disp_h,
                            float fx0, float fy0, float cx0, float cy0, float baseline, float* out_xyz, int cap);
int dvo_process_extra_points(const float* xyz, int n, float* out_xyz);
int dvo_insts_output(dvo_insts*, dvo_inst_obs* insts, int cap_insts, int* n_insts, dvo_feat* feats, int cap_feats, int* n_feats, double* points, int cap_points, int* n_points);
/* ProcessImage with the object branch (estimator.cpp:1562-1622,1653-1676); see inst_manager.h */
int dvo_estimator_process_dynamic(dvo_estimator*, const dvo_feat* feats, int n, double t, const dvo_inst_obs* insts, int n_insts, const dvo_feat* inst_feats,
                                  const double* points, dvo_be_state* out);
int dvo_fit_box_ransac(const double* pts, int n, const double* dims3, unsigned long long seed, double* out3);      /* vio_util.cpp:209-264, seeded (inst_manager.h) */
int dvo_fit_box_camera(const double* pts, int n, const double* dims3, double* out3);                                /* vio_util.cpp:274-332 */
void dvo_estimator_get_extrinsics(dvo_estimator*, double* ric18, double* tic6, double* td);      /* body.ric / tic / td after the last Double2vector */
int dvo_estimator_set_lines(dvo_estimator* e, const dvo_line_row* lines, int n);      /* frame.features.lines of the next process call */
int dvo_estimator_get_lines(dvo_estimator* e, dvo_line_landmark* out, int cap, int* n_out);
int dvo_estimator_get_instances(dvo_estimator*, dvo_inst_state* out, int cap, int* n_out, double* summary4);
/* InstanceManager::GetOutputInstInfo (estimator_insts.cpp:967-990) as FeatureTrack reads it (system/main.cpp:194,217-245): ids of the initialised, tracked instances that
 * were is_static at the last frame's snapshot (right behind PushBack), ascending */
int dvo_estimator_get_static_instances(dvo_estimator*, uint32_t* ids, int cap, int* n_out);

#ifdef __cplusplus
}
#endif
#endif
