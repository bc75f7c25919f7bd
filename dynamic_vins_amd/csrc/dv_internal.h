// dv_internal.h — shared declarations of the gfx950 implementation (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/dvins.h"

#define DV_LK_WIN 21          // cv::Size(21,21) at every reference call site
#define DV_MAX_LEVELS 4       // maxLevel = 3 -> 4 levels
#define DV_MAX_RADIUS 128     // largest disc radius (min_dist) supported by the mask stamper

// apron: pixels of BORDER_REFLECT_101 data stored around the w x h image (p points at pixel (0, 0) inside the padded buffer); 0 = none
struct DvLevel { uint8_t* p; int w, h, pitch, apron; };
#define DV_PYR_APRON 32      // >= the 27 px an LK tile can overhang the image, multiple of 4 (dword-aligned tile loads)
struct DvPyr   { DvLevel L[DV_MAX_LEVELS]; int levels; };   // levels = number of valid entries

// candidate record emitted by the Shi-Tomasi tile kernel
struct DvCand { unsigned long long key; float min_nb; int pad; };

// device-side tracker state (struct of arrays, capacity DV_MAX_FEATS)
struct DvTrackState {
    float2*   last_pts;      // positions in the previous frame
    float2*   curr_pts;      // positions in the current frame (compacted, sorted, + new corners)
    float2*   lk_pts;        // raw temporal-LK output (indexed like last_pts)
    uint8_t*  lk_status;
    uint32_t* ids;
    int32_t*  track_cnt;
    float2*   prev_un;       // undistorted position in the previous frame (prev_id_pts value)
    float2*   prev_run;      // right_prev_id_pts value
    uint8_t*  prev_rvalid;   // id present in right_prev_id_pts
    uint8_t*  tracked;       // 1 = id present in prev_id_pts
    float2*   right_pts;
    uint8_t*  right_status;
    int*      n_feat;        // device scalar: current number of features
    int*      n_tracked;     // device scalar: features that survived temporal LK
    uint32_t* next_id;       // InstFeat::global_id_count
    unsigned short* lk_order;   // scratch of dv_launch_lk_track: the points of the launch in position order
};

#define DV_GFTT_RULE_CPU 0
#define DV_GFTT_RULE_CUDA 1
struct GfttTileArgs {
    const uint8_t* img; int w, h, pitch;
    const uint8_t* in_mask; int mask_pitch;      // optional user mask (0 = excluded)
    const float2* disc_pts; const int* n_disc;   // tracked points whose discs are excluded (may be null)
    int radius; const uint8_t* hw;               // disc half widths hw[0..radius]
    const int* n_feat; int max_cnt;              // skip everything when max_cnt - *n_feat < min_new
    int min_new;
    float* eig_out; int eig_pitch;               // optional: write the eigenvalue image (elements)
    DvCand* cand; int cand_cap; int* n_cand; unsigned* max_ord;
    int rule;                                    // DV_GFTT_RULE_CPU (cv::goodFeaturesToTrack) | DV_GFTT_RULE_CUDA (cv::cuda::GoodFeaturesToTrackDetector: TrackImageNaive)
};

struct GfttSelectArgs {
    const DvCand* cand; const int* n_cand; int cand_cap; const unsigned* max_ord;
    int w, h; double quality; double min_dist;
    int max_n_host; const int* n_feat; int max_cnt; int min_new;   // tracker: max_n = max_cnt - *n_feat (needs >= min_new)
    float2* out_xy; int* n_out;                                    // operator-level output (dv_gftt), may be null
    DvTrackState tr; int has_tr;                                   // tracker epilogue: append corners, assign ids (1), or append and leave (old n_feat, accepted) in id_slot (2):
    int* err_flag;                                                 // the ids of several jobs drawing on ONE counter are handed out afterwards, in job order (dv_launch_gftt_assign_ids)
    int* id_slot;
    int rule;                                                      // as GfttTileArgs::rule
};

struct dv_ctx;
void dv_set_error(dv_ctx* ctx, const std::string& msg);

// ---- kernel launchers (defined in the .hip files) ----
void dv_launch_pyr_apron(const DvPyr& a, const DvPyr* b, hipStream_t s);      // fills the reflect-101 apron of every level (after the levels themselves)
void dv_launch_pyr_down2(const uint8_t* src0, const uint8_t* src1, int sw, int sh, int spitch,
                         uint8_t* dst0, uint8_t* dst1, int dpitch, uint8_t* copy0, uint8_t* copy1, int cpitch,
                         hipStream_t s, int rn_even = 0);      // rn_even: cuda::pyrDown's saturate_cast<uchar>(float) (round half to even) instead of cv::pyrDown's (sum + 128) >> 8
void dv_launch_bgr2gray(const uint8_t* src0, const uint8_t* src1, int w, int h, int spitch, uint8_t* dst0, uint8_t* dst1, int dpitch, hipStream_t s);
void dv_launch_remap(const uint8_t* src0, const uint8_t* src1, int w, int h, int spitch, int cn, int to_gray, const int16_t* m1_0, const uint16_t* m2_0,
                     const int16_t* m1_1, const uint16_t* m2_1, uint8_t* dst0, uint8_t* dst1, int dpitch, hipStream_t s);
void dv_launch_viode_mask(const uint8_t* seg, int w, int h, int spitch, const uint32_t* dyn_keys, int nkeys, uint8_t* merge, uint8_t* inv, int mpitch,
                          uint32_t* key_img, int32_t* boxes, hipStream_t s);
void dv_launch_lk_generic(const DvPyr& A, const DvPyr& B, const float2* pts_a, int n, int max_level, int iters,
                          double eps_sq, int use_initial, float2* pts_b, uint8_t* status, hipStream_t s);
// one job of dv_launch_lk_track_multi (device-resident table): FeatureTrackByLK of one object's points between its two pyramids
// one level step of one image pair for dv_launch_pyr_down_multi (src1 / dst1 may be null: single image)
struct DvPyrJob { const uint8_t* src0; const uint8_t* src1; uint8_t* dst0; uint8_t* dst1; int sw, sh, spitch, dw, dh, dpitch; uint8_t* cpy0; uint8_t* cpy1; int cpitch, pad_; };      // cpy (optional): the source tile's own pixels also go to a pitched copy (level 0 of the pyramid: frame read once)
void dv_launch_pyr_down_multi(const DvPyrJob* jobs_dev, int n_jobs, int max_dw, int max_dh, hipStream_t s);
void dv_launch_pyr_apron_multi(const DvPyr* pyrs_dev, int n_pyr, int max_levels, hipStream_t s);
// job-table forms of the tracker's single-workgroup / per-image stages (the front ends of a dv_batch group in shared launches: front_batch.hip)
struct DvCompactJob { DvTrackState tr; const uint8_t* in_mask; int mask_pitch, sort_by_cnt; int* n_cand; unsigned* max_ord; };
void dv_launch_compact_multi(const DvCompactJob* jobs_dev, int n_jobs, hipStream_t s);
struct DvFinalizeJob { DvTrackState tr; dv_cam cam0, cam1; int stereo, use_off; double dt; dv_feat* out; int* n_out; const int* err_in; int* err_out; float off_x, off_y; };      // use_off: the points are ROI-local, (off_x, off_y) = the box corner
void dv_launch_finalize_multi(const DvFinalizeJob* jobs_dev, int n_jobs, int n_max, hipStream_t s);
void dv_launch_gftt_tile_multi(const GfttTileArgs* tab_dev, int n_jobs, int w, int h, hipStream_t s);      // w x h = the largest image of the table; a smaller job leaves the tiles outside its own image idle
void dv_launch_gftt_assign_ids(const GfttSelectArgs* tab_dev, int n_jobs, hipStream_t s);                  // behind dv_launch_gftt_select_multi with has_tr == 2: ids in job order from the jobs' common counter
struct DvErodeJob { const uint8_t* src; uint8_t* tmp; uint8_t* dst; int w, h, spitch, tpitch, dpitch, k; };
void dv_launch_erode_multi(const DvErodeJob* jobs_dev, int n_jobs, int w_max, int h_max, hipStream_t s);
int  dv_launch_gftt_select_multi(const GfttSelectArgs* tab_dev, int n_jobs, hipStream_t s);
struct DvLkJob { DvPyr A, B; const float2* pts_a; const int* n_dev; float2* pts_b; uint8_t* status; float add_x, add_y; int use_add; uint32_t key; };      // key: the object's VIODE key (dv_launch_right_key_check)
// VIODE: a right-image point survives TrackRightByPad only where seg1 carries the object's key (instance_feature.cpp:263-268); one workgroup per job of the stereo LK table
void dv_launch_right_key_check(const DvLkJob* jobs_dev, int n_jobs, const uint32_t* key_img, int pitch_elems, int w, int h, hipStream_t s);
void dv_launch_lk_track_multi(const DvLkJob* jobs_dev, int n_jobs, int n_max, int flow_back, float dist_thresh, hipStream_t s);
void dv_launch_lk_track(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max,
                        int flow_back, float dist_thresh, float2* pts_b, uint8_t* status, hipStream_t s, unsigned short* order_scratch = nullptr);      // order_scratch ([n_max] ushort, device): track the points in position order, XCD-aware (lk.hip lk_order_kernel)
void dv_launch_lk_track_offset(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max, int flow_back,
                               float dist_thresh, float add_x, float add_y, float2* pts_b, uint8_t* status, hipStream_t s);
void dv_launch_finalize_offset(const DvTrackState& tr, const dv_cam& cam0, const dv_cam& cam1, int stereo, double dt, int n_max, double off_x, double off_y,
                               dv_feat* out, int* n_out, hipStream_t s);
// the reference's GPU tracker rule (lk_cuda.hip): A / B are pyramids built with cuda::pyrDown's rounding (dv_launch_pyr_down2(..., rn_even = 1)); level 0 is the frame
void dv_launch_lk_cuda_track(const DvPyr& A, const DvPyr& B, const float2* pts_a, const int* n_dev, int n_max, int flow_back, float dist_thresh, float2* pts_b, uint8_t* status, hipStream_t s);
void dv_launch_lk_cuda_generic(const DvPyr& A, const DvPyr& B, const float2* pts_a, int n, int max_level, int iters, int use_initial, float2* pts_b, uint8_t* status, hipStream_t s);
void dv_launch_gftt_tile(const GfttTileArgs& a, hipStream_t s);
// system/main.cpp:217-245: inv_mask(y0 + r, x0 + c) = 255 where roi_mask(r, c) >= 1 (a static instance's pixels become background); roi_mask: w x h bytes in PINNED host memory, read in place
void dv_launch_unmask(uint8_t* inv_mask, int pitch, int W, int H, int x0, int y0, int w, int h, const uint8_t* roi_mask, hipStream_t s);
hipError_t dv_copy_async(void* dst, const void* src, size_t bytes, hipStream_t s);      // copy.hip: device <-> PINNED host (or device <-> device) as a kernel on s — no copy engine in the per-frame path
int  dv_launch_gftt_select(const GfttSelectArgs& a, hipStream_t s);
void dv_launch_compact(const DvTrackState& tr, const uint8_t* in_mask, int mask_pitch, int sort_by_cnt, int* n_cand,
                       unsigned* max_ord, hipStream_t s);
void dv_launch_finalize(const DvTrackState& tr, const dv_cam& cam0, const dv_cam& cam1, int stereo, double dt, int n_max,
                        dv_feat* out, int* n_out, hipStream_t s, const int* err_in = nullptr, int* err_out = nullptr);      // out / n_out / err_out may be pinned host memory; *err_out = *err_in by thread 0
void dv_launch_circle_mask(uint8_t* mask, int w, int h, int pitch, const float2* pts, int n, int radius, const uint8_t* hw,
                           hipStream_t s);
void dv_launch_erode(const uint8_t* src, int w, int h, int spitch, int k, uint8_t* tmp, int tpitch, uint8_t* dst, int dpitch,
                     hipStream_t s);
void dv_launch_lift(const dv_cam& cam, const float2* in, int n, double off_x, double off_y, float2* out, hipStream_t s);
// the extra-point pipeline of dynamic mode (extra_points.hip): one job = one visible object
#define DV_XP_CAP 3200       // points per object after sampling: step = max(sqrt(0.8 rows cols / 1000), 2) bounds the grid by ~3130 nodes for images up to 1280 wide
struct DvExtraJob { const uint8_t* mask; int mask_pitch, cols, rows, box_x, box_y, step; double* out; int* n_out; };      // out: 3 * DV_XP_CAP doubles, n_out: count (device or pinned host memory)
struct DvExtraArgs { const float* disp; int disp_pitch /* elements */, disp_w, disp_h; float fx0, fy0, cx0, cy0, baseline; int* err_flag; int stage /* 1: stop after the sampling */; uint8_t* pool /* dv_extra_points_scratch_bytes(n_jobs) of device scratch */; };
#define DV_XP_SCRATCH_BYTES ((size_t)DV_XP_CAP * (16 + 16 + 1 + 8) + 256 + 1024)      // per object: two point sets, keep flags, two label sets, control words
size_t dv_extra_points_scratch_bytes(int n_jobs);
int dv_launch_extra_points(const DvExtraJob* jobs_dev, int n_jobs, const DvExtraArgs& a, hipStream_t s);
int dv_extra_points_step(int rows, int cols);
