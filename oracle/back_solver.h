// back_solver.h — CPU ORACLE (test infrastructure, not the product): restatement of ceres::Solve as the
// reference configures it (estimator/estimator.cpp:296-314): trust region, TRADITIONAL_DOGLEG,
// DENSE_SCHUR, Jacobi scaling, monotonic steps, HuberLoss corrector, local parameterisations.
// Ceres 1.14.0 is an un-vendored dependency (README.md:17); the algorithm below follows its published
// sources (trust_region_minimizer.cc, dogleg_strategy.cc, corrector.cc, schur_complement_solver.cc) as
// summarised in SURVEY.md App. A.3.  PARITY UNPINNED for this piece.  Canonical choices:
//   S1  Schur e-blocks = the inverse-depth blocks only (Ceres picks a maximal independent set that may
//       also contain speed-bias blocks; the eliminated linear system and its solution are the same).
//   S2  The wall-clock budget (max_solver_time) is disabled; iteration cap = max_num_iterations (Q18).
// The normal equations are formed blockwise (H = J^T J, g = J^T f) instead of materialising J; every
// quantity the dogleg needs (column norms, J*v products, model cost change) is a function of H and g.
#pragma once
extern int g_var_radius;      // dvo_set_variant("radius", .): front_oracle.cpp
#include <functional>
#include <memory>
#include "la.h"

namespace obe {
using namespace ola;

enum BlockKind { kPlain = 0, kPose = 1, kPosePlaneImu = 2, kPosePlaneVo = 3, kLineOrth = 4 };
extern "C" void dvo_line_plus(const double* orth4, const double* delta4, double* out4);      // LineOrthParameterization::Plus (obj_factors.cpp)

struct ParamBlock {
    double* data = nullptr; int size = 0; int kind = kPlain; bool constant = false; bool is_e = false;
    int local() const { return (size == 7 && kind != kPlain) ? 6 : size; }      // pose blocks: tangent size 6; a plain 7-block has no parameterisation
    int col = -1;   // offset in the tangent vector of the reduced ordering (assigned by Solve)
};

struct CostFunction {
    int nres = 0; std::vector<int> sizes;
    virtual ~CostFunction() = default;
    // J[k]: nres x sizes[k] row-major, or null
    virtual void Evaluate(const double* const* par, double* res, double** J) const = 0;
};

enum LossKind { kNoLoss = 0, kHuber1 = 1, kCauchy1 = 2 };

struct ResidualBlock { std::shared_ptr<CostFunction> f; int loss = kNoLoss; std::vector<ParamBlock*> blocks; };

struct Problem {
    std::vector<std::unique_ptr<ParamBlock>> params;
    std::vector<ResidualBlock> residuals;
    ParamBlock* find(double* p) { for (auto& b : params) if (b->data == p) return b.get(); return nullptr; }
    ParamBlock* AddParameterBlock(double* p, int size, int kind = kPlain, bool is_e = false) {
        if (auto* b = find(p)) return b;
        params.emplace_back(new ParamBlock{ p, size, kind, false, is_e });
        return params.back().get();
    }
    void SetConstant(double* p) { find(p)->constant = true; }
    void AddResidualBlock(std::shared_ptr<CostFunction> f, int loss, std::vector<double*> ps) {
        ResidualBlock rb; rb.f = std::move(f); rb.loss = loss;
        for (size_t i = 0; i < ps.size(); ++i) rb.blocks.push_back(AddParameterBlock(ps[i], rb.f->sizes[i]));
        residuals.push_back(std::move(rb));
    }
};

inline void loss_eval(int kind, double s, double rho[3]) {     // ceres::HuberLoss(1.0) / CauchyLoss(1.0)
    if (kind == kHuber1) {
        if (s > 1.0) { const double r = std::sqrt(s); rho[0] = 2 * r - 1; rho[1] = std::max(std::numeric_limits<double>::min(), 1.0 / r); rho[2] = -rho[1] / (2 * s); }
        else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
    } else if (kind == kCauchy1) {
        const double sum = 1 + s, inv = 1 / sum;
        rho[0] = std::log(sum); rho[1] = std::max(std::numeric_limits<double>::min(), inv); rho[2] = -(inv * inv);
    } else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
}

// ceres Corrector (== the first-party copy at factor/marginalization_factor.cpp:54-78)
inline void correct(int loss, int nres, double* res, std::vector<std::vector<double>>& J, const std::vector<int>& cols, double* cost) {
    double sq = 0; for (int i = 0; i < nres; ++i) sq += res[i] * res[i];
    if (loss == kNoLoss) { if (cost) *cost = 0.5 * sq; return; }
    double rho[3]; loss_eval(loss, sq, rho);
    if (cost) *cost = 0.5 * rho[0];
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else { const double D = 1.0 + 2.0 * sq * rho[2] / rho[1]; const double alpha = 1.0 - std::sqrt(D); residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq; }
    for (size_t k = 0; k < J.size(); ++k) {
        if (J[k].empty()) continue;
        const int c = cols[k];
        std::vector<double> rtJ(c, 0.0);
        for (int i = 0; i < nres; ++i) for (int j = 0; j < c; ++j) rtJ[j] += res[i] * J[k][i * c + j];
        for (int i = 0; i < nres; ++i) for (int j = 0; j < c; ++j) J[k][i * c + j] = sqrt_rho1 * (J[k][i * c + j] - alpha_sq_norm * res[i] * rtJ[j]);
    }
    for (int i = 0; i < nres; ++i) res[i] *= residual_scaling;
}

// x (+) delta for one block (PoseLocalParameterization::Plus, factor/pose_local_parameterization.cpp:26-103)
inline void plus(const ParamBlock& b, const double* x, const double* d, double* out) {
    if (b.kind == kPlain) { for (int i = 0; i < b.size; ++i) out[i] = x[i] + d[i]; return; }
    if (b.kind == kLineOrth) { dvo_line_plus(x, d, out); return; }      // 4 global, 4 local parameters, ComputeJacobian = I
    V3 dp(d[0], d[1], d[2]);
    if (b.kind == kPosePlaneImu) dp.z = 0;
    if (b.kind == kPosePlaneVo) dp.y = 0;
    out[0] = x[0] + dp.x; out[1] = x[1] + dp.y; out[2] = x[2] + dp.z;
    Q q(x[6], x[3], x[4], x[5]);
    Q r = (q * deltaQ(V3(d[3], d[4], d[5]))).normalized();
    out[3] = r.x; out[4] = r.y; out[5] = r.z; out[6] = r.w;
}

struct SolveOptions { int max_num_iterations = 8; double xnorm2_extra = 0; };     // xnorm2_extra: squared norm of blocks in ceres' x that the flat problem does not carry (dvo_ba_problem::x_norm2_extra)
struct SolveSummary {
    int iterations = 0, successful = 0; double initial_cost = 0, final_cost = 0; int termination = 0;   // 0 max-iter, 1 converged, 2 failure
    std::vector<double> cost_trace;
};

class Solver {
public:
    Problem& P; std::vector<ParamBlock*> vars; int np = 0, ne = 0, N = 0;
    // normal equations in the tangent space, UNSCALED
    Mat Hpp; std::vector<double> Hpe; std::vector<double> Hee, g;   // Hpe: np x ne (dense; oracle sizes are small)
    explicit Solver(Problem& p) : P(p) {
        for (auto& b : P.params) if (!b->constant && !b->is_e) { b->col = np; np += b->local(); vars.push_back(b.get()); }
        for (auto& b : P.params) if (!b->constant && b->is_e) { b->col = np + ne; ne += b->local(); vars.push_back(b.get()); }
        N = np + ne;
    }
    // evaluates cost (and, if build, the normal equations) at the current parameter values
    double evaluate(bool build) {
        if (build) { Hpp = Mat(np, np); Hpe.assign((size_t)np * ne, 0.0); Hee.assign(ne, 0.0); g.assign(N, 0.0); }
        double cost = 0;
        std::vector<const double*> par; std::vector<double*> Jp; std::vector<std::vector<double>> J; std::vector<int> cols; std::vector<double> res;
        for (auto& rb : P.residuals) {
            const int nb = (int)rb.blocks.size(), nr = rb.f->nres;
            par.resize(nb); Jp.assign(nb, nullptr); J.assign(nb, {}); cols.resize(nb); res.assign(nr, 0.0);
            for (int k = 0; k < nb; ++k) {
                par[k] = rb.blocks[k]->data; cols[k] = rb.blocks[k]->size;
                if (build && !rb.blocks[k]->constant) { J[k].assign((size_t)nr * cols[k], 0.0); Jp[k] = J[k].data(); }
            }
            rb.f->Evaluate(par.data(), res.data(), build ? Jp.data() : nullptr);
            double c; correct(rb.loss, nr, res.data(), J, cols, &c);
            cost += c;
            if (!build) continue;
            // local parameterisation: pose blocks keep the first 6 columns (ComputeJacobian = [I6; 0])
            for (int a = 0; a < nb; ++a) {
                if (J[a].empty()) continue;
                const ParamBlock& ba = *rb.blocks[a]; const int la = ba.local(), ca = cols[a];
                for (int i = 0; i < la; ++i) { double s = 0; for (int r = 0; r < nr; ++r) s += J[a][r * ca + i] * res[r]; g[ba.col + i] += s; }
                for (int b = a; b < nb; ++b) {
                    if (J[b].empty()) continue;
                    const ParamBlock& bb = *rb.blocks[b]; const int lb = bb.local(), cb = cols[b];
                    for (int i = 0; i < la; ++i) for (int j = 0; j < lb; ++j) {
                        double s = 0; for (int r = 0; r < nr; ++r) s += J[a][r * ca + i] * J[b][r * cb + j];
                        add(ba.col + i, bb.col + j, s, a == b);
                    }
                }
            }
        }
        return cost;
    }
    void add(int i, int j, double s, bool same_block) {
        auto put = [&](int a, int b) {
            if (a < np && b < np) Hpp(a, b) += s;
            else if (a < np && b >= np) Hpe[(size_t)a * ne + (b - np)] += s;
            else if (a >= np && b >= np) { assert(a == b && "e-blocks must be independent"); Hee[a - np] += s; }
        };
        put(i, j);
        if (!same_block) { if (j < np && i < np) Hpp(j, i) += s; else if (j < np && i >= np) Hpe[(size_t)j * ne + (i - np)] += s; }
    }
    double hdiag(int i) const { return i < np ? Hpp(i, i) : Hee[i - np]; }
    // y = H x (full block Hessian, unscaled)
    void hmul(const std::vector<double>& x, std::vector<double>& y) const {
        y.assign(N, 0.0);
        for (int i = 0; i < np; ++i) { double s = 0; for (int j = 0; j < np; ++j) s += Hpp(i, j) * x[j]; for (int e = 0; e < ne; ++e) s += Hpe[(size_t)i * ne + e] * x[np + e]; y[i] = s; }
        for (int e = 0; e < ne; ++e) { double s = Hee[e] * x[np + e]; for (int i = 0; i < np; ++i) s += Hpe[(size_t)i * ne + e] * x[i]; y[np + e] = s; }
    }
    void gather(std::vector<double>& x) const { x.clear(); for (auto* b : vars) x.insert(x.end(), b->data, b->data + b->size); }
    void scatter(const std::vector<double>& x) const { size_t o = 0; for (auto* b : vars) { std::memcpy(b->data, x.data() + o, sizeof(double) * b->size); o += b->size; } }
    void apply(const std::vector<double>& x, const std::vector<double>& delta, std::vector<double>& out) const {
        out.resize(x.size()); size_t o = 0;
        for (auto* b : vars) { plus(*b, x.data() + o, delta.data() + b->col, out.data() + o); o += b->size; }
    }

    SolveSummary solve(const SolveOptions& opt) {
        SolveSummary sum;
        std::vector<double> x, cand, scale(N), diag(N), grad(N), gn(N), step(N), delta(N), tmp, tmp2;
        gather(x);
        double x_cost = evaluate(true);
        sum.initial_cost = x_cost; sum.cost_trace.push_back(x_cost);
        // jacobi scaling, fixed for the whole solve: 1 / (1 + sqrt(||col||^2))
        for (int i = 0; i < N; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(hdiag(i)));
        double radius = 1e4, mu = 1e-8; const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
        bool reuse = false; double alpha = 0, dogleg_norm = 0; int invalid = 0;
        double x_norm = opt.xnorm2_extra; for (double v : x) x_norm += v * v; x_norm = std::sqrt(x_norm);
        auto grad_max = [&]() { double m = 0; for (int i = 0; i < N; ++i) m = std::max(m, std::fabs(g[i])); return m; };   // plain-Euclidean blocks; pose blocks use Plus-projected gradient in Ceres, max-norm equal to first order
        if (grad_max() <= 1e-10) { sum.termination = 1; sum.final_cost = x_cost; return sum; }
        for (int it = 1;; ++it) {
            if (it > opt.max_num_iterations) { sum.termination = 0; break; }
            sum.iterations = it;
            bool step_valid = true;
            if (!reuse) {
                reuse = true;
                for (int i = 0; i < N; ++i) { double v = hdiag(i) * scale[i] * scale[i]; v = std::min(std::max(v, 1e-6), 1e32); diag[i] = std::sqrt(v); }
                for (int i = 0; i < N; ++i) grad[i] = g[i] * scale[i] / diag[i];
                // Cauchy point
                tmp.resize(N); for (int i = 0; i < N; ++i) tmp[i] = grad[i] / diag[i] * scale[i];
                hmul(tmp, tmp2);
                double JgJg = 0, gg = 0; for (int i = 0; i < N; ++i) { JgJg += tmp[i] * tmp2[i]; gg += grad[i] * grad[i]; }
                alpha = gg / JgJg;
                // Gauss-Newton step with mu regularisation
                bool ok = false;
                while (true) {
                    ok = gauss_newton(scale, diag, mu, gn);
                    if (ok) break;
                    mu *= mu_inc;
                    if (mu > max_mu) break;
                }
                if (ok) for (int i = 0; i < N; ++i) gn[i] *= -diag[i];
                else step_valid = false;
            }
            double model_cost_change = 0;
            if (step_valid) {
                // traditional dogleg
                double gnorm = 0, gnn = 0; for (int i = 0; i < N; ++i) { gnorm += grad[i] * grad[i]; gnn += gn[i] * gn[i]; } gnorm = std::sqrt(gnorm); gnn = std::sqrt(gnn);
                if (gnn <= radius) { step = gn; dogleg_norm = gnn; }
                else if (gnorm * alpha >= radius) { for (int i = 0; i < N; ++i) step[i] = -(radius / gnorm) * grad[i]; dogleg_norm = radius; }
                else {
                    double gdot = 0; for (int i = 0; i < N; ++i) gdot += grad[i] * gn[i];
                    const double b_dot_a = -alpha * gdot, a2 = std::pow(alpha * gnorm, 2.0), bma2 = a2 - 2 * b_dot_a + std::pow(gnn, 2);
                    const double c = b_dot_a - a2, d = std::sqrt(c * c + bma2 * (std::pow(radius, 2.0) - a2));
                    const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
                    double n2 = 0; for (int i = 0; i < N; ++i) { step[i] = (-alpha * (1.0 - beta)) * grad[i] + beta * gn[i]; n2 += step[i] * step[i]; }
                    dogleg_norm = std::sqrt(n2);
                }
                for (int i = 0; i < N; ++i) step[i] /= diag[i];
                // model_cost_change = -(J s)^T (f + J s / 2) with J the jacobi-scaled Jacobian
                tmp.resize(N); for (int i = 0; i < N; ++i) tmp[i] = step[i] * scale[i];
                hmul(tmp, tmp2);
                double sg = 0, sHs = 0; for (int i = 0; i < N; ++i) { sg += tmp[i] * g[i]; sHs += tmp[i] * tmp2[i]; }
                model_cost_change = -(sg + 0.5 * sHs);
                step_valid = model_cost_change > 0.0;
                if (step_valid) { delta = tmp; invalid = 0; }
            }
            if (!step_valid) {
                if (++invalid >= 5) { sum.termination = 2; break; }
                mu *= mu_inc; reuse = false;            // StepIsInvalid
                sum.cost_trace.push_back(x_cost);
                continue;
            }
            apply(x, delta, cand);
            scatter(cand);
            const double cand_cost = evaluate(false);
            // parameter tolerance
            double sn = 0; for (size_t i = 0; i < x.size(); ++i) sn += (x[i] - cand[i]) * (x[i] - cand[i]); sn = std::sqrt(sn);
            if (sn <= 1e-8 * (x_norm + 1e-8)) { scatter(x); sum.termination = 1; break; }
            // function tolerance
            if (std::fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { scatter(x); sum.termination = 1; break; }
            const double rel = (x_cost - cand_cost) / model_cost_change;
            if (rel > 1e-3) {
                x = cand; x_norm = opt.xnorm2_extra; for (double v : x) x_norm += v * v; x_norm = std::sqrt(x_norm);
                x_cost = evaluate(true);    // == cand_cost; rebuilds H, g at the new point
                sum.successful++;
                if (rel < 0.25) radius *= 0.5;
                if (rel > 0.75) radius = g_var_radius == 1 ? 3.0 * radius : std::max(radius, 3.0 * dogleg_norm);      // (variant 1: sensitivity reading, dvo.h)
                mu = std::max(min_mu, 2.0 * mu / mu_inc);
                reuse = false;
                sum.cost_trace.push_back(x_cost);
                if (grad_max() <= 1e-10) { sum.termination = 1; break; }
            } else {
                scatter(x);
                radius *= 0.5; reuse = true;
                sum.cost_trace.push_back(x_cost);
                if (radius < 1e-32) { sum.termination = 1; break; }
            }
        }
        scatter(x);
        sum.final_cost = x_cost;
        return sum;
    }

    // solves (Hs + mu diag^2) y = gs with Hs = S H S, gs = S g via Schur elimination of the e-part
    bool gauss_newton(const std::vector<double>& scale, const std::vector<double>& diag, double mu, std::vector<double>& y) {
        Mat S(np, np); std::vector<double> rhs(np), einv(ne);
        for (int e = 0; e < ne; ++e) {
            const int k = np + e;
            const double d = Hee[e] * scale[k] * scale[k] + mu * diag[k] * diag[k];
            if (!(d > 0) || !std::isfinite(d)) return false;
            einv[e] = 1.0 / d;
        }
        for (int i = 0; i < np; ++i) {
            for (int j = 0; j < np; ++j) S(i, j) = Hpp(i, j) * scale[i] * scale[j];
            S(i, i) += mu * diag[i] * diag[i];
            rhs[i] = g[i] * scale[i];
        }
        for (int e = 0; e < ne; ++e) {
            const int k = np + e; const double ge = g[k] * scale[k];
            std::vector<int> nz; nz.reserve(64);
            for (int i = 0; i < np; ++i) if (Hpe[(size_t)i * ne + e] != 0.0) nz.push_back(i);
            for (int i : nz) {
                const double wi = Hpe[(size_t)i * ne + e] * scale[i] * scale[k];
                rhs[i] -= wi * einv[e] * ge;
                for (int j : nz) S(i, j) -= wi * einv[e] * Hpe[(size_t)j * ne + e] * scale[j] * scale[k];
            }
        }
        Mat L;
        if (np > 0 && !cholesky(S, L)) return false;
        if (np > 0) chol_solve(L, rhs);
        y.assign(N, 0.0);
        for (int i = 0; i < np; ++i) y[i] = rhs[i];
        for (int e = 0; e < ne; ++e) {
            const int k = np + e; double s = g[k] * scale[k];
            for (int i = 0; i < np; ++i) { const double w = Hpe[(size_t)i * ne + e]; if (w != 0.0) s -= w * scale[i] * scale[k] * y[i]; }
            y[k] = s * einv[e];
        }
        for (int i = 0; i < N; ++i) if (!std::isfinite(y[i])) return false;
        return true;
    }
};

}  // namespace obe
