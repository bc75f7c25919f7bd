// obj_solve.cpp — CPU ORACLE (test infrastructure, not the product): the two auxiliary solves with a block-diagonal Hessian (dvo_obj_solve,
// dvo_line_solve, the latter restating Estimator::OptimizationWithOnlyLine, estimator/estimator.cpp:345-395): the per-frame object solve of dynamic mode
// (numeric part), restated from
//   InstanceManager::Optimization                 estimator/estimator_insts.cpp:772-807   (DENSE_SCHUR + DOGLEG, HuberLoss(1.0))
//   InstanceManager::AddInstanceParameterBlock    estimator/estimator_insts.cpp:989-1010  (para_state[i]: Pose / PoseConstraint local parameterisation)
//   InstanceManager::AddResidualBlockForInstOpt   estimator/estimator_insts.cpp:1018-1245 (the residual blocks that are not commented out:
//        BoxDimsFactor + BoxOrientationFactor per frame with a 3-D box, BoxEncloseStereoPointFactor per triangulated point)
// on top of the generic ceres::Solve restatement in back_solver.h and the factor bodies in obj_factors.cpp.
// The problem is built the way the reference builds it — including the body pose blocks, which are ordinary (7-wide,
// non-constant) parameter blocks of this ceres::Problem whose Jacobian the orientation factor leaves at zero; blocks
// without a residual are left out, as ceres' preprocessor removes them.  PARITY UNPINNED (Ceres is un-vendored, dvo.h).
#include <cstring>
#include <memory>
#include <vector>
#include "dvo.h"
#include "back_solver.h"

using namespace obe;

namespace {

struct EncloseCost : CostFunction {
    double pw[3], dims[3];
    EncloseCost(const double* p, const double* d) { nres = 3; sizes = { 7 }; std::memcpy(pw, p, 24); std::memcpy(dims, d, 24); }
    void Evaluate(const double* const* par, double* res, double** J) const override { dvo_box_enclose_eval(pw, dims, par, res, J); }
};
struct DimsCost : CostFunction {
    double dims[3];
    explicit DimsCost(const double* d) { nres = 1; sizes = { 3 }; std::memcpy(dims, d, 24); }
    void Evaluate(const double* const* par, double* res, double** J) const override { dvo_box_dims_eval(dims, par, res, J); }
};
struct OrientationCost : CostFunction {
    double Rc[9], Rb[9];
    OrientationCost(const double* rc, const double* rb) { nres = 3; sizes = { 7, 7 }; std::memcpy(Rc, rc, 72); std::memcpy(Rb, rb, 72); }
    void Evaluate(const double* const* par, double* res, double** J) const override { dvo_box_orientation_eval(Rc, Rb, par, res, J); }
};

// lineProjectionFactor on (pose, ex_pose, line) — estimator/factor/line_projection_factor.cpp:24-159 through dvo_line_eval
struct LineCost : CostFunction {
    double obs[4], si[4];
    LineCost(const double* o, const double* s) { nres = 2; sizes = { 7, 7, 4 }; std::memcpy(obs, o, 32); std::memcpy(si, s, 32); }
    void Evaluate(const double* const* par, double* res, double** J) const override { dvo_line_eval(obs, si, par, res, J); }
};

}  // namespace

// Estimator::OptimizationWithOnlyLine (estimator/estimator.cpp:345-395) + AddLineResidualBlock (:222-253): pose and extrinsic blocks are added
// with PoseLocalParameterization and set constant, one LineOrthParameterization block per line, CauchyLoss(1.0), DENSE_SCHUR + DOGLEG.
extern "C" int dvo_line_solve(dvo_line_problem* P, dvo_ba_summary* S) {
    std::vector<double> pose(P->pose, P->pose + 77), ex(P->ex_pose, P->ex_pose + 7);
    Problem prob;
    for (int i = 0; i < 11; ++i) { prob.AddParameterBlock(pose.data() + 7 * i, 7, kPose); prob.SetConstant(pose.data() + 7 * i); }
    prob.AddParameterBlock(ex.data(), 7, kPose); prob.SetConstant(ex.data());
    for (int i = 0; i < P->n_obs; ++i) {
        const dvo_line_obs& ob = P->obs[i];
        prob.AddParameterBlock(P->orth + 4 * (size_t)ob.line, 4, kLineOrth);
        prob.AddResidualBlock(std::make_shared<LineCost>(ob.obs, P->sqrt_info), kCauchy1, { pose.data() + 7 * ob.frame, ex.data(), P->orth + 4 * (size_t)ob.line });
    }
    Solver solver(prob);
    SolveOptions opt; opt.max_num_iterations = P->max_iters;
    const SolveSummary sum = solver.solve(opt);
    S->iterations = sum.iterations; S->successful = sum.successful; S->termination = sum.termination; S->slots = 0;
    S->initial_cost = sum.initial_cost; S->final_cost = sum.final_cost;
    return 0;
}

extern "C" int dvo_obj_solve(dvo_obj_problem* P, dvo_ba_summary* S) {
    const int n_obj = P->n_obj;
    std::vector<double> body(P->body_pose, P->body_pose + 77);          // ceres may write parameter blocks; the caller's array is const
    std::vector<double> dims0(P->dims, P->dims + 3 * (size_t)n_obj);    // inst.box3d->dims, passed to the point factor by value
    Problem prob;
    const int pose_kind = P->plane_kind == 0 ? kPose : (P->plane_kind == 1 ? kPosePlaneImu : kPosePlaneVo);
    auto state = [&](int o, int f) { return P->state + 7 * ((size_t)o * 11 + f); };
    // AddResidualBlockForInstOpt iterates objects, then frames (boxes), then landmarks; the order only affects summation order
    for (int i = 0; i < P->n_boxes; ++i) {
        const dvo_obj_box& b = P->boxes[i];
        prob.AddParameterBlock(state(b.obj, b.frame), 7, pose_kind);
        prob.AddResidualBlock(std::make_shared<DimsCost>(b.dims), kHuber1, { P->dims + 3 * (size_t)b.obj });
        prob.AddResidualBlock(std::make_shared<OrientationCost>(b.R_cioi, P->R_bc), kNoLoss, { body.data() + 7 * b.frame, state(b.obj, b.frame) });
    }
    // variant "obj_point_order" 1 (sensitivity only, dvo.h): the point blocks join the problem in REVERSE order — same parameter blocks in the same order, same residual
    // blocks, a different (equally valid) summation order of J^T J and J^T r.  What that is worth over a long dynamic run: tests/tools/obj_sensitivity.py
    const bool rev = dvo_get_variant("obj_point_order") == 1;
    if (rev) for (int i = 0; i < P->n_points; ++i) prob.AddParameterBlock(state(P->points[i].obj, P->points[i].frame), 7, pose_kind);
    for (int k = 0; k < P->n_points; ++k) {
        const dvo_obj_point& p = P->points[rev ? P->n_points - 1 - k : k];
        prob.AddParameterBlock(state(p.obj, p.frame), 7, pose_kind);
        prob.AddResidualBlock(std::make_shared<EncloseCost>(p.p_w, dims0.data() + 3 * (size_t)p.obj), kHuber1, { state(p.obj, p.frame) });
    }
    Solver solver(prob);
    SolveOptions opt; opt.max_num_iterations = P->max_iters;
    const SolveSummary sum = solver.solve(opt);
    S->iterations = sum.iterations; S->successful = sum.successful; S->termination = sum.termination; S->slots = 0;
    S->initial_cost = sum.initial_cost; S->final_cost = sum.final_cost;
    return 0;
}
