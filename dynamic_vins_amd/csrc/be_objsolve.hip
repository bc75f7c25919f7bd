// be_objsolve.hip — the per-frame solve of the dynamic objects on gfx950 (numeric part): replaces
// ceres::Solve inside InstanceManager::Optimization (estimator/estimator_insts.cpp:772-807) for the problem built by
// AddInstanceParameterBlock / AddResidualBlockForInstOpt (estimator_insts.cpp:989-1245): per object 11 pose blocks and
// one dims block; BoxDimsFactor + BoxOrientationFactor per 3-D detection, BoxEncloseStereoPointFactor per triangulated
// point; DENSE_SCHUR + DOGLEG with HuberLoss(1.0), Jacobi scaling, monotonic steps.
//
// Structure the kernel is built around: every residual block touches exactly one variable block (the body pose enters
// the orientation factor with a zero Jacobian, sic), so J^T J is BLOCK DIAGONAL — 6x6 per (object, frame), 3x3 per
// object — while the dogleg trust region couples all blocks through a handful of scalars (|g|, |gn|, g.gn, the model
// decrease, the candidate cost).  The whole trust-region loop therefore runs inside ONE persistent workgroup launch:
//   eval      8 lanes per variable block stride over its points (coalesced SoA reads), lane-reduce the 3x3 + 3 + cost
//             partials with wave shuffles, lane 0 adds the orientation factor and writes the block's H | g
//   solve     one thread per variable block: Jacobi-scaled, mu-regularised 6x6 Cholesky in registers
//   scalars   workgroup reductions (wave shuffle + 8 LDS partials) that every thread reads back identically, so the
//             trust-region bookkeeping is replicated in registers and control flow stays uniform
// The trust-region loop itself is the generic block-diagonal solver of bd_solve.h; this file supplies the problem (evaluation, plus).
// No host round trip inside the solve; one H2D of the packed problem, one launch, one D2H of the states.
// The problem is K objects x (11 + 1) blocks and a few thousand points: latency-bound, neither HBM nor MFMA matter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "dv_ctx.h"
#include "be_math.h"
#include "be_obj_dev.h"
#include "bd_solve.h"
#include "dev_once.h"

using namespace be;

namespace {
using namespace bd;

constexpr int OS_NF = 11;                   // kWinSize + 1 pose blocks per object

struct ObjProb {
    int n_obj, nblk, npts, plane_kind;
    const double* dims0;                    // n_obj x 3: dims at entry (constants of the point factor)
    const double* body;                     // 11 x 7
    const double* rbc;                      // 9
    const double* pt;                       // SoA x | y | z, npts each, sorted by block
    const int* pt_start;                    // nblk + 1
    const double* box_R;                    // nblk x 9
    const double* box_dims;                 // nblk x 3
    const unsigned char* has_box;           // nblk

    __device__ int xdim(int v) const { return v < nblk ? 7 : 3; }
    __device__ int plus(int v, const double* x, const double* d, double* o) const {
        if (v < nblk) { pose_plus(x, d, plane_kind, o); return 7; }
        for (int i = 0; i < 3; ++i) o[i] = x[i] + d[i];
        return 3;
    }
    template <bool BUILD>
    __device__ void eval(const BdArgs& ba, const double* __restrict__ xb, double* __restrict__ Hb, double& cost, double& gmax) const {
        const ObjProb& a = *this;
    cost = 0; gmax = 0;
    const int tid = threadIdx.x, grp = tid / BD_GROUP, j = tid % BD_GROUP;
    for (int b = grp; b < a.nblk; b += BD_THREADS / BD_GROUP) {
        if (!ba.active[b]) continue;                           // uniform over the 8 lanes
        const int k0 = a.pt_start[b], k1 = a.pt_start[b + 1];
        const double* xp = xb + 7 * b;
        const d3 P = P3(xp); const quat q = Q4(xp);
        const double* d0 = a.dims0 + 3 * (b / OS_NF);
        double acc[10];                                       // H_pp (6, packed lower) | g_p (3) | cost
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = 0;
        for (int k = k0 + j; k < k1; k += BD_GROUP) {
            double r[3], J[9];
            box_enclose_dev(mk3(a.pt[k], a.pt[a.npts + k], a.pt[2 * a.npts + k]), d0, P, q, r, J);
            double rho0, s;
            huber1(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, s);
            acc[9] += 0.5 * rho0;
            if (BUILD) {
#pragma unroll
                for (int k2 = 0; k2 < 9; ++k2) J[k2] *= s;
                r[0] *= s; r[1] *= s; r[2] *= s;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[6 + c] += J[c] * r[0] + J[3 + c] * r[1] + J[6 + c] * r[2];
#pragma unroll
                    for (int c2 = 0; c2 <= c; ++c2) acc[tri(c, c2)] += J[c] * J[c2] + J[3 + c] * J[3 + c2] + J[6 + c] * J[6 + c2];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            double v = acc[k];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            acc[k] = v;
        }
        if (j == 0) {
            double hr[6] = { 0, 0, 0, 0, 0, 0 }, gr[3] = { 0, 0, 0 };
            if (a.has_box[b]) {
                m33 Rc, Rb;
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rc.m[k] = a.box_R[9 * b + k]; Rb.m[k] = a.rbc[k]; }
                double r[3], J[9];
                box_orientation_dev(Rc, Rb, Q4(a.body + 7 * (b % OS_NF)), q, r, J);
                acc[9] += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);          // no loss function on this factor
                if (BUILD) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        gr[c] = J[c] * r[0] + J[3 + c] * r[1] + J[6 + c] * r[2];
#pragma unroll
                        for (int c2 = 0; c2 <= c; ++c2) hr[tri(c, c2)] = J[c] * J[c2] + J[3 + c] * J[3 + c2] + J[6 + c] * J[6 + c2];
                    }
                }
            }
            cost += acc[9];
            if (BUILD) {
                double* h = Hb + (size_t)BD_HSTRIDE * b;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int c2 = 0; c2 <= c; ++c2) { h[tri(c, c2)] = acc[tri(c, c2)]; h[tri(3 + c, 3 + c2)] = hr[tri(c, c2)]; }
#pragma unroll
                    for (int c2 = 0; c2 < 3; ++c2) h[tri(3 + c, c2)] = 0.0;
                    h[21 + c] = acc[6 + c]; h[24 + c] = gr[c];
                    gmax = fmax(gmax, fmax(fabs(acc[6 + c]), fabs(gr[c])));
                }
            }
        }
    }
    // dims blocks: at most 11 one-dimensional residuals each, one lane per object
    for (int o = tid; o < a.n_obj; o += BD_THREADS) {
        const int v = a.nblk + o;
        if (!ba.active[v]) continue;
        const d3 box = P3(xb + 7 * v);
        double h6[6] = { 0, 0, 0, 0, 0, 0 }, g3[3] = { 0, 0, 0 }, c = 0;
        for (int f = 0; f < OS_NF; ++f) {
            const int b = o * OS_NF + f;
            if (!a.has_box[b]) continue;
            double r, J[3];
            box_dims_dev(box, P3(a.box_dims + 3 * b), r, J);
            double rho0, s;
            huber1(r * r, rho0, s);
            c += 0.5 * rho0;
            if (BUILD) {
                r *= s;
#pragma unroll
                for (int k = 0; k < 3; ++k) J[k] *= s;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    g3[k] += J[k] * r;
#pragma unroll
                    for (int k2 = 0; k2 <= k; ++k2) h6[tri(k, k2)] += J[k] * J[k2];
                }
            }
        }
        cost += c;
        if (BUILD) {
            double* h = Hb + (size_t)BD_HSTRIDE * v;
#pragma unroll
            for (int k = 0; k < BD_HSTRIDE; ++k) h[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) h[k] = h6[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) { h[21 + k] = g3[k]; gmax = fmax(gmax, fabs(g3[k])); }
        }
    }
}

};

}  // namespace


// internal two-phase form: the estimator's dynamic branch enqueues the object solve on its own stream + scratch buffer (pinned staging for both
// directions, no host synchronisation) so that it runs beside the window solve; _end waits for its event and unpacks
int be_obj_solve_begin(dv_ctx* ctx, dv_obj_problem* P, hipStream_t s, DevBuf& scratch, ObjPending& pend) {
    if (!ctx) return -1;
    if (pend.active) DV_FAIL("dv_obj_solve: previous object solve not collected");
    if (!P) DV_FAIL("dv_obj_solve: null argument");
    if (P->n_obj <= 0 || P->n_boxes < 0 || P->n_points < 0 || P->max_iters < 0 || !P->state || !P->dims || !P->body_pose) DV_FAIL("dv_obj_solve: bad problem");
    if ((P->n_boxes > 0 && !P->boxes) || (P->n_points > 0 && !P->points)) DV_FAIL("dv_obj_solve: null factor list");
    if (P->plane_kind < 0 || P->plane_kind > 2) DV_FAIL("dv_obj_solve: bad plane_kind");
    DV_CHECK(hipSetDevice(ctx->cfg.device));
    const int n_obj = P->n_obj, nblk = n_obj * OS_NF, V = nblk + n_obj, npts = P->n_points;

    // host side: bucket the factors by variable block (stable counting sort), pack one upload
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_x0 = carve(8 * 7 * (size_t)V), o_dims0 = carve(8 * 3 * (size_t)n_obj), o_body = carve(8 * 7 * OS_NF), o_rbc = carve(8 * 9),
                 o_pt = carve(8 * 3 * (size_t)std::max(npts, 1)), o_start = carve(4 * ((size_t)nblk + 1)), o_boxR = carve(8 * 9 * (size_t)nblk),
                 o_boxd = carve(8 * 3 * (size_t)nblk), o_hasb = carve((size_t)nblk), o_act = carve((size_t)V);
    const size_t up_bytes = off;
    const size_t o_x1 = carve(8 * 7 * (size_t)V), o_H0 = carve(8 * BD_HSTRIDE * (size_t)V), o_H1 = carve(8 * BD_HSTRIDE * (size_t)V), o_vec = carve(8 * 30 * (size_t)V), o_out = carve(64);
    const size_t dl_bytes = 8 * 7 * (size_t)V + 64;
    if (pend.pinned_bytes < up_bytes + dl_bytes) {
        if (pend.pinned) (void)hipHostFree(pend.pinned);
        pend.pinned = nullptr; pend.pinned_bytes = 0;
        // with a floor (4 MB ~ 16 objects x 11 frames with ~150 k points): the object problem GROWS over the first frames after the objects are initialised, and every
        // hipHostFree / hipHostMalloc pair is a multi-millisecond call that holds runtime locks the tracker thread's launches wait for (seen as 7 ms frames at the
        // start of the timed region of the dynamic bench line)
        const size_t want = std::max<size_t>(2 * (up_bytes + dl_bytes), (size_t)4 << 20);
        DV_CHECK(hipHostMalloc(&pend.pinned, want, hipHostMallocDefault));
        pend.pinned_bytes = want;
    }
    if (!pend.ev) DV_CHECK(hipEventCreateWithFlags(&pend.ev, hipEventDisableTiming));
    struct HostView { uint8_t* p; uint8_t* data() { return p; } } host{ (uint8_t*)pend.pinned };
    std::memset(host.p, 0, up_bytes);
    double* hx = (double*)(host.data() + o_x0); double* hd0 = (double*)(host.data() + o_dims0); double* hpt = (double*)(host.data() + o_pt);
    int* hstart = (int*)(host.data() + o_start); double* hR = (double*)(host.data() + o_boxR); double* hbd = (double*)(host.data() + o_boxd);
    uint8_t* hhas = host.data() + o_hasb; uint8_t* hact = host.data() + o_act;
    for (int b = 0; b < nblk; ++b) memcpy(hx + 7 * b, P->state + 7 * (size_t)b, 56);
    for (int o = 0; o < n_obj; ++o) { memcpy(hx + 7 * (size_t)(nblk + o), P->dims + 3 * o, 24); memcpy(hd0 + 3 * o, P->dims + 3 * o, 24); }
    memcpy(host.data() + o_body, P->body_pose, 8 * 7 * OS_NF);
    memcpy(host.data() + o_rbc, P->R_bc, 72);
    bool frame_has_box[OS_NF] = {};
    for (int i = 0; i < P->n_boxes; ++i) {
        const dv_obj_box& bx = P->boxes[i];
        if (bx.obj < 0 || bx.obj >= n_obj || bx.frame < 0 || bx.frame >= OS_NF) DV_FAIL("dv_obj_solve: box index out of range");
        const int b = bx.obj * OS_NF + bx.frame;
        if (hhas[b]) DV_FAIL("dv_obj_solve: more than one box for an (object, frame)");
        hhas[b] = 1; hact[b] = 1; hact[nblk + bx.obj] = 1; frame_has_box[bx.frame] = true;
        memcpy(hR + 9 * (size_t)b, bx.R_cioi, 72); memcpy(hbd + 3 * (size_t)b, bx.dims, 24);
    }
    for (int i = 0; i < npts; ++i) {
        const dv_obj_point& p = P->points[i];
        if (p.obj < 0 || p.obj >= n_obj || p.frame < 0 || p.frame >= OS_NF) DV_FAIL("dv_obj_solve: point index out of range");
        hstart[p.obj * OS_NF + p.frame + 1]++;
    }
    for (int b = 0; b < nblk; ++b) { if (hstart[b + 1]) hact[b] = 1; hstart[b + 1] += hstart[b]; }
    {
        std::vector<int> fill(hstart, hstart + nblk);
        for (int i = 0; i < npts; ++i) {
            const dv_obj_point& p = P->points[i];
            const int k = fill[p.obj * OS_NF + p.frame]++;
            hpt[k] = p.p_w[0]; hpt[npts + k] = p.p_w[1]; hpt[2 * (size_t)npts + k] = p.p_w[2];
        }
    }
    double xc = 0;
    for (int f = 0; f < OS_NF; ++f) if (frame_has_box[f]) for (int k = 0; k < 7; ++k) xc += P->body_pose[7 * f + k] * P->body_pose[7 * f + k];

    if (scratch.ensure(std::max<size_t>(off, (size_t)8 << 20)) != hipSuccess) DV_FAIL("dv_obj_solve: out of device memory");      // (floor: a hipFree + hipMalloc per growth step is a device-wide synchronisation)
    uint8_t* base = (uint8_t*)scratch.p;
    DV_CHECK(dv_copy_async(base, host.data(), up_bytes, s));      // pinned -> HBM by a kernel on the object stream (copy.hip)
    BdArgs a{};          // (x1 <- x0 and the zeroing of H | vec are the kernel's own first phase: two enqueued operations less in front of it)
    a.V = V; a.max_iters = P->max_iters;
    a.x0 = (double*)(base + o_x0); a.x1 = (double*)(base + o_x1); a.H0 = (double*)(base + o_H0); a.H1 = (double*)(base + o_H1); a.vec = (double*)(base + o_vec);
    a.active = base + o_act; a.xnorm2_const = xc; a.out = (double*)(base + o_out);
    ObjProb pr{};
    pr.n_obj = n_obj; pr.nblk = nblk; pr.npts = npts; pr.plane_kind = P->plane_kind;
    pr.dims0 = (const double*)(base + o_dims0); pr.body = (const double*)(base + o_body); pr.rbc = (const double*)(base + o_rbc); pr.pt = (const double*)(base + o_pt);
    pr.pt_start = (const int*)(base + o_start); pr.box_R = (const double*)(base + o_boxR); pr.box_dims = (const double*)(base + o_boxd); pr.has_box = base + o_hasb;
    // the solved states and the summary reach the pinned buffer with the kernel's completion (written by the kernel itself over PCIe): no download copies behind it
    uint8_t* dl = (uint8_t*)pend.pinned + up_bytes;
    a.h_x = (double*)dl; a.h_out = (double*)(dl + 8 * 7 * (size_t)V);
    const size_t lds = bd_lds_bytes(V);
    a.lds = lds <= BD_LDS_MAX ? 1 : 0;          // up to 10 objects the whole working set lives in LDS (bd_solve.h); larger problems keep it in HBM
    {
        static DevOnce once;
        if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(bd_solve_kernel<ObjProb>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BD_LDS_MAX) != hipSuccess ? 1 : 0; })) DV_FAIL("dv_obj_solve: cannot set the dynamic LDS size");
        StageScope sc(ctx, "obj_solve", s);
        hipLaunchKernelGGL(bd_solve_kernel<ObjProb>, dim3(1), dim3(BD_THREADS), a.lds ? lds : 0, s, pr, a);
    }
    DV_CHECK(hipGetLastError());
    DV_CHECK(hipEventRecord(pend.ev, s));
    pend.active = true; pend.V = V; pend.nblk = nblk; pend.n_obj = n_obj; pend.up_bytes = up_bytes; pend.stream = s;
    return 0;
}

int be_obj_solve_prepare(dv_ctx* ctx, DevBuf& scratch, ObjPending& pend) {
    if (!pend.pinned) {
        const size_t want = (size_t)4 << 20;      // the floor of be_obj_solve_begin
        DV_CHECK(hipHostMalloc(&pend.pinned, want, hipHostMallocDefault));
        pend.pinned_bytes = want;
    }
    if (!pend.ev) DV_CHECK(hipEventCreateWithFlags(&pend.ev, hipEventDisableTiming));
    if (scratch.ensure((size_t)8 << 20) != hipSuccess) DV_FAIL("dv_obj_solve: out of device memory");
    static DevOnce once;
    if (once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(bd_solve_kernel<ObjProb>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BD_LDS_MAX) != hipSuccess ? 1 : 0; })) DV_FAIL("dv_obj_solve: cannot set the dynamic LDS size");
    return 0;
}

int be_obj_solve_end(dv_ctx* ctx, dv_obj_problem* P, dv_ba_summary* summary, ObjPending& pend) {
    if (!ctx) return -1;
    if (!pend.active) DV_FAIL("dv_obj_solve: nothing to collect");
    pend.active = false;
    DV_CHECK(hipEventSynchronize(pend.ev));
    if (ctx->timing) { DV_CHECK(hipStreamSynchronize(pend.stream)); dv_harvest_timers(ctx, pend.stream); }
    const double* hxo = (const double*)((const uint8_t*)pend.pinned + pend.up_bytes); const double* hout = hxo + 7 * (size_t)pend.V;
    for (int b = 0; b < pend.nblk; ++b) memcpy(P->state + 7 * (size_t)b, hxo + 7 * (size_t)b, 56);
    for (int o = 0; o < pend.n_obj; ++o) memcpy(P->dims + 3 * o, hxo + 7 * (size_t)(pend.nblk + o), 24);
    if (summary) {
        summary->iterations = (int)hout[0]; summary->successful = (int)hout[1]; summary->termination = (int)hout[2]; summary->slots = 0;
        summary->initial_cost = hout[3]; summary->final_cost = hout[4];
    }
    return 0;
}

extern "C" int dv_obj_solve(dv_ctx* ctx, dv_obj_problem* P, dv_ba_summary* summary) {
    if (!ctx) return -1;
    if (!summary) DV_FAIL("dv_obj_solve: null argument");
    if (be_obj_solve_begin(ctx, P, ctx->be_stream, ctx->s1, ctx->obj_op_pend)) return -1;
    return be_obj_solve_end(ctx, P, summary, ctx->obj_op_pend);
}
