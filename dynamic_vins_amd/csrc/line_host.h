// line_host.h — host side of line mode in the back end (product code): line landmarks of FeatureManager and their geometry, on flat vectors.
// Mirrors
//   FeatureManager::{AddFeatureCheckParallax (line part),TriangulateLineMono,GetLineFeatureCount,GetLineOrthVector,SetLineOrth,RemoveLineOutlier}
//   and the line halves of RemoveBackShiftDepth / RemoveBack / RemoveFront                                  estimator/feature_manager.cpp:124-160,339-356,392-560,611-778
//   TriangulateOneLine                                                                                     estimator/vio_util.cpp:447-561
//   plk_to_orth / orth_to_plk / pi_from_ppp / pipi_plk / plk_to_pose / plk_from_pose / LineReprojectionError / LineTrimming   line_detector/line_geometry.cpp:75-296
// The numeric parts with data-parallel work — lineProjectionFactor evaluation and the line-only dogleg solve — run on the GPU (be_linesolve.hip);
// what is here is O(lines x observations) scalar geometry per frame.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "be_math.h"
#include "../../include/dvins.h"

namespace dvl {
using namespace be;

struct Plk { d3 n, v; };                              // Pluecker line: normal of the plane through the origin, direction
struct Pl4 { d3 n; double d; };                       // plane n . x + d = 0

inline Pl4 plane_ppp(d3 x1, d3 x2, d3 x3) { Pl4 p; p.n = cross(x1 - x3, x2 - x3); p.d = -dot(x3, cross(x1, x2)); return p; }      // pi_from_ppp
inline Plk plk_from_planes(const Pl4& a, const Pl4& b) {                                                                     // pipi_plk: dual Pluecker matrix a b^T - b a^T
    Plk l;
    l.n = a.n * b.d - b.n * a.d;                                       // (dp03, dp13, dp23)
    l.v = mk3(-(a.n.y * b.n.z - b.n.y * a.n.z), a.n.x * b.n.z - b.n.x * a.n.z, -(a.n.x * b.n.y - b.n.x * a.n.y));      // (-dp12, dp02, -dp01)
    return l;
}
inline Plk plk_to_pose(const Plk& w, const m33& Rcw, d3 tcw) { Plk c; c.v = mul(Rcw, w.v); c.n = mul(Rcw, w.n) + mul(mul(skew(tcw), Rcw), w.v); return c; }
inline Plk plk_from_pose(const Plk& c, const m33& Rcw, d3 tcw) { const m33 Rwc = tr(Rcw); return plk_to_pose(c, Rwc, -mul(Rwc, tcw)); }
inline void plk_to_orth(const Plk& l, double o[4]) {
    const double nn = norm(l.n), vn = norm(l.v);
    const d3 u1 = l.n / nn, u2 = l.v / vn, u3 = cross(u1, u2);
    o[0] = atan2(u2.z, u3.z); o[1] = asin(-u1.z); o[2] = atan2(u1.y, u1.x);
    const double wn = sqrt(nn * nn + vn * vn);
    o[3] = asin(vn / wn);
}
inline Plk orth_to_plk(const double o[4]) {
    const double s1 = sin(o[0]), c1 = cos(o[0]), s2 = sin(o[1]), c2 = cos(o[1]), s3 = sin(o[2]), c3 = cos(o[2]);
    const d3 u1 = mk3(c2 * c3, c2 * s3, -s2), u2 = mk3(s1 * s2 * c3 - c1 * s3, s1 * s2 * s3 + c1 * c3, s1 * c2);      // first two columns of R
    Plk l; l.n = u1 * cos(o[3]); l.v = u2 * sin(o[3]); return l;
}
// LineTrimming: the two 3-D end points (camera frame) where the planes through the observed end points, perpendicular to the image line, cut the line
inline bool line_trimming(const Plk& l, const double obs[4], d3& e1, d3& e2) {
    const d3 p11 = mk3(obs[0], obs[1], 1.0), p21 = mk3(obs[2], obs[3], 1.0);
    const d3 c = cross(p11, p21);
    const double ln = sqrt(c.x * c.x + c.y * c.y), lx = c.x / ln, ly = c.y / ln;
    const d3 p12 = mk3(p11.x + lx, p11.y + ly, 1.0), p22 = mk3(p21.x + lx, p21.y + ly, 1.0), cam = mk3(0, 0, 0);
    auto cut = [&](const Pl4& pi, d3& e) {            // Lc * pi with Lc = [skew(n) v; -v^T 0], de-homogenised
        const d3 h = mul(skew(l.n), pi.n) + l.v * pi.d; const double w = -dot(l.v, pi.n);
        e = h / w; return e.z;
    };
    const double z1 = cut(plane_ppp(cam, p11, p12), e1), z2 = cut(plane_ppp(cam, p21, p22), e2);
    return z1 >= 0 && z2 >= 0;
}
inline double line_reproj_error(const double obs[4], const m33& Rwc, d3 twc, const Plk& lw) {
    const Plk lc = plk_from_pose(lw, Rwc, twc);
    const double sql = sqrt(lc.n.x * lc.n.x + lc.n.y * lc.n.y);
    const d3 nc = lc.n / sql;
    return (fabs(dot(nc, mk3(obs[0], obs[1], 1.0))) + fabs(dot(nc, mk3(obs[2], obs[3], 1.0)))) / 2.0;
}

struct LObs { double l[4], r[4]; bool stereo; };       // LineFeature
struct LLm {                                           // LineLandmark
    int id = 0, start = 0; std::vector<LObs> obs; bool tri = false; Plk plk{}; d3 ptw1, ptw2; int end() const { return start + (int)obs.size() - 1; }
};

struct LineMgr {
    std::vector<LLm> lms; int min_obs = 5; static constexpr int kW = 10;
    void clear() { lms.clear(); }
    bool usable(const LLm& l) const { return (int)l.obs.size() >= min_obs && l.start < kW - 2 && l.tri; }      // the filter of AddLineResidualBlock / GetLineOrthVector / SetLineOrth / RemoveLineOutlier
    int count() const { int c = 0; for (auto& l : lms) if (usable(l)) ++c; return c; }
    // AddFeatureCheckParallax, line part: FeatureBackground::lines is a std::map keyed by line id
    void add(int fc, const dv_line_row* rows, int n) {
        std::vector<const dv_line_row*> order(n);
        for (int i = 0; i < n; ++i) order[i] = &rows[i];
        std::stable_sort(order.begin(), order.end(), [](const dv_line_row* a, const dv_line_row* b) { return a->id < b->id; });
        for (const dv_line_row* r : order) {
            LObs o; o.stereo = r->has_right != 0;
            for (int k = 0; k < 4; ++k) { o.l[k] = r->left[k]; o.r[k] = o.stereo ? r->right[k] : 0.0; }
            LLm* hit = nullptr;
            for (auto& l : lms) if (l.id == (int)r->id) { hit = &l; break; }
            if (!hit) { lms.emplace_back(); lms.back().id = (int)r->id; lms.back().start = fc; hit = &lms.back(); }
            hit->obs.push_back(o);
        }
    }
    // TriangulateLineMono + TriangulateOneLine: the pair of observation planes with the widest angle between them defines the line
    void triangulate(const m33* Rs, const d3* Ps, const m33& ric, d3 tic) {
        for (auto& L : lms) {
            if ((int)L.obs.size() < min_obs || L.tri) continue;
            const int i = L.start;
            const d3 t0 = Ps[i] + mul(Rs[i], tic); const m33 R0 = mul(Rs[i], ric), R0t = tr(R0);
            double min_cos = 1.0; Pl4 pii{}; d3 ni = mk3(0, 0, 0), tij = mk3(0, 0, 0); m33 Rij = eye3(); const double* obsj = nullptr;
            for (size_t k = 0; k < L.obs.size(); ++k) {
                const double* o = L.obs[k].l; const int j = i + (int)k;
                if (k == 0) { pii = plane_ppp(mk3(o[0], o[1], 1), mk3(o[2], o[3], 1), mk3(0, 0, 0)); ni = pii.n / norm(pii.n); continue; }
                const d3 t1 = Ps[j] + mul(Rs[j], tic); const m33 R1 = mul(Rs[j], ric);
                const d3 t = mul(R0t, t1 - t0); const m33 R = mul(R0t, R1);
                const d3 p3 = mul(R, mk3(o[0], o[1], 1)) + t, p4 = mul(R, mk3(o[2], o[3], 1)) + t;
                const Pl4 pij = plane_ppp(p3, p4, t);
                const d3 nj = pij.n / norm(pij.n);
                const double c = dot(ni, nj);
                if (c < min_cos) { min_cos = c; tij = t; Rij = R; obsj = o; }
            }
            if (min_cos > 0.998 || !obsj) continue;
            const d3 p3 = mul(Rij, mk3(obsj[0], obsj[1], 1)) + tij, p4 = mul(Rij, mk3(obsj[2], obsj[3], 1)) + tij;
            const Plk plk = plk_from_planes(pii, plane_ppp(p3, p4, tij));
            d3 e1, e2;
            if (!line_trimming(plk, L.obs[0].l, e1, e2) || norm(e1 - e2) > 10.0) continue;
            L.ptw1 = mul(Rs[i], mul(ric, e1) + tic) + Ps[i]; L.ptw2 = mul(Rs[i], mul(ric, e2) + tic) + Ps[i];
            L.plk = plk; L.tri = true;
        }
    }
    // GetLineOrthVector: world-frame orthonormal representation of every usable line
    void get_orth(const m33* Rs, const d3* Ps, const m33& ric, d3 tic, std::vector<double>& out) const {
        out.clear();
        for (auto& L : lms) {
            if (!usable(L)) continue;
            const d3 twc = Ps[L.start] + mul(Rs[L.start], tic); const m33 Rwc = mul(Rs[L.start], ric);
            double o[4]; plk_to_orth(plk_to_pose(L.plk, Rwc, twc), o);
            out.insert(out.end(), o, o + 4);
        }
    }
    void set_orth(const m33* Rs, const d3* Ps, const m33& ric, d3 tic, const double* x) {
        int k = -1;
        for (auto& L : lms) {
            if (!usable(L)) continue;
            const Plk lw = orth_to_plk(x + 4 * (++k));
            const d3 twc = Ps[L.start] + mul(Rs[L.start], tic); const m33 Rwc = mul(Rs[L.start], ric);
            L.plk = plk_from_pose(lw, Rwc, twc);
        }
    }
    // the observations of AddLineResidualBlock (estimator.cpp:224-253): line index, window frame, left observation
    void observations(std::vector<dv_line_obs>& out) const {
        out.clear();
        int k = -1;
        for (auto& L : lms) {
            if (!usable(L)) continue;
            ++k;
            for (size_t q = 0; q < L.obs.size(); ++q) { dv_line_obs ob{}; ob.line = k; ob.frame = L.start + (int)q; for (int c = 0; c < 4; ++c) ob.obs[c] = L.obs[q].l[c]; out.push_back(ob); }
        }
    }
    void remove_outliers(const m33* Rs, const d3* Ps, const m33& ric, d3 tic) {      // RemoveLineOutlier (feature_manager.cpp:518-560)
        lms.erase(std::remove_if(lms.begin(), lms.end(), [&](const LLm& L) {
            if (!usable(L)) return false;
            d3 e1, e2;
            if (!line_trimming(L.plk, L.obs[0].l, e1, e2) || norm(e1 - e2) > 10) return true;
            const d3 twc = Ps[L.start] + mul(Rs[L.start], tic); const m33 Rwc = mul(Rs[L.start], ric);
            const Plk lw = plk_to_pose(L.plk, Rwc, twc);
            double worst = 0;
            for (size_t q = 0; q < L.obs.size(); ++q) {
                const int j = L.start + (int)q;
                const double err = line_reproj_error(L.obs[q].l, mul(Rs[j], ric), Ps[j] + mul(Rs[j], tic), lw);
                if (worst < err) worst = err;
            }
            return worst > 3.0 / 500.0;
        }), lms.end());
    }
    // line halves of RemoveBackShiftDepth (marg_R / marg_P, new_R / new_P are CAMERA poses), RemoveBack, RemoveFront
    void remove_back_shift(const m33& marg_R, d3 marg_P, const m33& new_R, d3 new_P) {
        for (auto& L : lms) {
            if (L.start != 0) { L.start--; continue; }
            L.obs.erase(L.obs.begin());
            if (L.obs.size() < 2) { L.id = -1; continue; }
            const m33 Rji = mul(tr(new_R), marg_R); const d3 tji = mul(tr(new_R), marg_P - new_P);
            L.plk = plk_to_pose(L.plk, Rji, tji);
        }
        lms.erase(std::remove_if(lms.begin(), lms.end(), [](const LLm& L) { return L.id < 0; }), lms.end());
    }
    void remove_back() {
        for (auto& L : lms) { if (L.start != 0) L.start--; else { L.obs.erase(L.obs.begin()); if (L.obs.empty()) L.id = -1; } }
        lms.erase(std::remove_if(lms.begin(), lms.end(), [](const LLm& L) { return L.id < 0; }), lms.end());
    }
    void remove_front(int fc) {
        for (auto& L : lms) {
            if (L.start == fc) { L.start--; continue; }
            if (L.end() < fc - 1) continue;
            L.obs.erase(L.obs.begin() + (kW - 1 - L.start));
            if (L.obs.empty()) L.id = -1;
        }
        lms.erase(std::remove_if(lms.begin(), lms.end(), [](const LLm& L) { return L.id < 0; }), lms.end());
    }
};

}  // namespace dvl
