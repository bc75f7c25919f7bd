// extra_points.cpp — CPU ORACLE (test infrastructure, NOT the product) of the object "extra point" pipeline of dynamic mode (second half):
//   InstFeat::DetectExtraPoints              front_end/instance_feature.cpp:413-461   strided disparity sampling inside the ROI mask -> camera-frame 3-D points
//   InstsFeatManager::ProcessExtraPoints     front_end/dynamic_tracker.cpp:159-340    pcl::RadiusOutlierRemoval(0.5 m, 10) + pcl::EuclideanClusterExtraction(1 m, 10, 25000),
//                                                                                     the first (= largest) cluster replaces extra_points3d
// First-party arithmetic follows the cited lines type by type (float camera parameters, cam_s: utils/camera_model.h:36-38; int pixel indices; float results widened
// to the Vec3d the reference stores).  The PCL 1.8 pieces are NOT under /root/reference and the reference has no test that pins them: restated from the published
// algorithms — PARITY UNPINNED, like the OpenCV / Ceres pieces (dvo.h):
//   * RadiusOutlierRemoval::applyFilterIndices on a dense cloud: k-nearest search with k = min_pts + 1 (the query point included); a point survives iff it HAS that
//     many neighbours and the farthest of them is not beyond the radius: !(r^2 < d_k^2).  Equivalent to "at least 11 points (itself included) with d^2 <= 0.25".
//     (The NaN-tolerant branch of the same function uses a strict radius search; the two differ only at d^2 == r^2 exactly.)  Output keeps the input order.
//   * KdTreeFLANN distances: flann::L2_Simple<float> — ((dx dx) + dy dy) + dz dz accumulated in float in x, y, z order; the search is exact (checks = -1, eps = 0),
//     so a brute-force scan with the same distance returns the same neighbour SETS; nothing here depends on the order inside a neighbour list.
//   * extractEuclideanClusters: region growing over radiusSearch(1.0) — FLANN's radius test is strict (d^2 < r^2) —, seeds in index order, so cluster k is the k-th
//     connected component by lowest member index; kept if 10 <= size <= 25000, member indices sorted ascending; EuclideanClusterExtraction::extract then sorts the
//     clusters by size, largest first (std::sort over reverse iterators with comparePointClusters: for the handful of clusters of an object an insertion sort, which
//     leaves equal sizes in discovery order).  cluster_indices[0] is therefore the largest component, the earliest-found one among equals.
// Quirk kept on purpose (DESIGN.md quirk ledger): the reference runs this on a second thread that reads roi->mask_cv while the tracking thread erodes that mask in
// place (dynamic_tracker.cpp:378 vs :425) — a data race.  The sampling is the first thing the new thread does and the erosion comes after the per-object optical
// flow, so the canonical reading is the UN-eroded mask; dvo_insts_track calls this before its erosion.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "dvo.h"

extern "C" {

// mask: rows x cols (roi->mask_cv, CV_8UC1, > 0 = object); (box_x, box_y) = box2d->rect.tl(); disp: CV_32F, disp_w x disp_h.  out_xyz: cap triples (float, as computed).
int dvo_detect_extra_points(const uint8_t* mask, int cols, int rows, int box_x, int box_y, const float* disp, int disp_w, int disp_h,
                            float fx0, float fy0, float cx0, float cy0, float baseline, float* out_xyz, int cap) {
    if (!mask || !disp || rows <= 0 || cols <= 0) return 0;                          // !roi || disp.empty()
    const float N_max = 1000.f;
    const int step = (int)std::max(std::sqrt(0.8 * rows * cols / N_max), 2.);        // double arithmetic, truncated (:421-422)
    int n = 0;
    for (int i = 0; i < rows; i += step) {
        for (int j = 0; j < cols; j += step) {
            if (mask[(size_t)i * cols + j] <= 0.5) continue;
            const int r = (int)((float)i + (float)box_y);                            // int + Rect2f::tl().y -> float -> int
            const int c = (int)((float)j + (float)box_x);
            if (r < 0 || r >= disp_h || c < 0 || c >= disp_w) continue;              // (cv::Mat::at is unchecked in release builds; the box lies inside the image)
            const float disparity = disp[(size_t)r * disp_w + c];
            if (disparity <= 0) continue;
            if (disparity != disparity) continue;
            const float depth = fx0 * baseline / disparity;
            if (depth <= 0.1 || depth > 100) continue;                               // float vs double literal: 0.1 is the double 0.1
            const float x_3d = ((float)c - cx0) * depth / fx0;
            const float y_3d = ((float)r - cy0) * depth / fy0;
            if (n < cap) { out_xyz[3 * n] = x_3d; out_xyz[3 * n + 1] = y_3d; out_xyz[3 * n + 2] = depth; }
            ++n;
        }
    }
    return n;
}

static inline float l2_simple(const float* a, const float* b) {      // flann::L2_Simple<float>
    float result = 0.f;
    for (int i = 0; i < 3; ++i) { const float diff = a[i] - b[i]; result += diff * diff; }
    return result;
}

// xyz: n float triples (the pcl::PointXYZ cloud EigenToPclXYZ builds).  out_xyz: up to n triples.  returns the size of the result (0: extra_points3d stays cleared).
int dvo_process_extra_points(const float* xyz, int n, float* out_xyz) {
    if (n <= 0) return 0;
    // ---- RadiusOutlierRemoval(radius 0.5, min neighbours 10) ----
    const int mean_k = 10 + 1;
    const double nn_dists_max = 0.5 * 0.5;
    std::vector<int> kept;
    std::vector<float> d2((size_t)n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) d2[j] = l2_simple(xyz + 3 * i, xyz + 3 * j);
        const int k = std::min(mean_k, n);
        bool chk = false;
        if (k == mean_k) {
            std::vector<float> s(d2);
            std::nth_element(s.begin(), s.begin() + (k - 1), s.end());              // nearestKSearch: the k-th smallest squared distance (the query itself is the 1st)
            chk = !(nn_dists_max < (double)s[k - 1]);
        }
        if (chk) kept.push_back(i);
    }
    if (kept.empty() || kept.size() < 5) return 0;                                   // dynamic_tracker.cpp:287-289
    const int m = (int)kept.size();
    std::vector<float> f((size_t)3 * m);
    for (int i = 0; i < m; ++i) std::memcpy(&f[3 * (size_t)i], xyz + 3 * (size_t)kept[i], 12);
    // ---- EuclideanClusterExtraction(tolerance 1.0, min 10, max 25000) ----
    const float r2 = (float)(1.0 * 1.0);
    std::vector<char> processed((size_t)m, 0);
    std::vector<std::vector<int>> clusters;
    for (int i = 0; i < m; ++i) {
        if (processed[i]) continue;
        std::vector<int> seed_queue; size_t sq_idx = 0;
        seed_queue.push_back(i); processed[i] = 1;
        while (sq_idx < seed_queue.size()) {
            const float* q = &f[3 * (size_t)seed_queue[sq_idx]];
            for (int j = 0; j < m; ++j) {
                if (processed[j]) continue;
                if (l2_simple(q, &f[3 * (size_t)j]) < r2) { seed_queue.push_back(j); processed[j] = 1; }
            }
            ++sq_idx;
        }
        if (seed_queue.size() >= 10 && seed_queue.size() <= 25000) { std::sort(seed_queue.begin(), seed_queue.end()); clusters.push_back(seed_queue); }
    }
    if (clusters.empty()) return 0;                                                  // :301-303
    size_t best = 0;
    for (size_t c = 1; c < clusters.size(); ++c) if (clusters[c].size() > clusters[best].size()) best = c;      // largest first; equal sizes keep discovery order
    const std::vector<int>& idx = clusters[best];
    for (size_t k = 0; k < idx.size(); ++k) std::memcpy(out_xyz + 3 * k, &f[3 * (size_t)idx[k]], 12);
    return (int)idx.size();
}

}  // extern "C"
