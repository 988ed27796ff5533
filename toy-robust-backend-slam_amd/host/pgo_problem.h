// The slice of the ceres API that DCS-ceres/main.cpp:66-164 uses, re-hosted on libpgo.so:
//   Problem::AddResidualBlock(cost, loss, p1, p2)   main.cpp:99,114,128,137,148
//   Problem::SetParameterBlockConstant(p)           main.cpp:153
//   Solver::Options / Solver::Summary / Solve()     main.cpp:154-164
// Parameter blocks are identified by their double* (Node::p), as in Ceres, and are updated in place.
#ifndef PGO_HOST_PROBLEM_H_
#define PGO_HOST_PROBLEM_H_

#include <cstdio>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "ceres_error.h"
#include "pgo.h"

namespace pgo {

struct LossFunction {
  virtual ~LossFunction() {}
  virtual double huber_delta() const = 0;
};
struct HuberLoss : LossFunction {  // ceres::HuberLoss(a), main.cpp:68
  explicit HuberLoss(double a) : a_(a) {}
  double huber_delta() const override { return a_; }
  double a_;
};

enum LinearSolverType { SPARSE_NORMAL_CHOLESKY, BLOCK_JACOBI_PCG };

class Problem {
 public:
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* p1, double* p2) {
    std::unique_ptr<CostFunction> own(cost);  // TAKE_OWNERSHIP, as Ceres does by default
    if (p1 == p2) throw std::invalid_argument("duplicate parameter block in a residual block");
    ia_.push_back(block(p1));
    ib_.push_back(block(p2));
    meas_.insert(meas_.end(), {cost->dx, cost->dy, cost->dtheta});
    if (cost->kind == CostFunction::SWITCHABLE || cost->kind == CostFunction::SWITCH_PRIOR)
      throw std::invalid_argument("switchable blocks take the (P1, P2, S) / (S) overloads");
    const bool dcs = cost->kind == CostFunction::DCS;
    kind_.push_back(dcs ? PGO_EDGE_CLOSURE : PGO_EDGE_ODOMETRY);
    switch_.push_back(nullptr);
    any_dcs_ = any_dcs_ || dcs;
    const double d = loss ? loss->huber_delta() : 0.0;
    if (!ia_.empty() && ia_.size() > 1 && d != delta_) mixed_loss_ = true;
    delta_ = d;
  }
  // SwitchableClosureResidue: (P1, P2, S)  (reference main.cpp:122,143)
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* p1, double* p2, double* s) {
    std::unique_ptr<CostFunction> own(cost);
    if (cost->kind != CostFunction::SWITCHABLE) throw std::invalid_argument("three parameter blocks: SwitchableClosureResidue only");
    if (p1 == p2) throw std::invalid_argument("duplicate parameter block in a residual block");
    ia_.push_back(block(p1));
    ib_.push_back(block(p2));
    meas_.insert(meas_.end(), {cost->dx, cost->dy, cost->dtheta});
    kind_.push_back(PGO_EDGE_CLOSURE);
    switch_.push_back(s);
    any_sc_ = true;
    const double d = loss ? loss->huber_delta() : 0.0;
    if (ia_.size() > 1 && d != delta_) mixed_loss_ = true;
    delta_ = d;
  }
  // SwitchPriorResidue: (S), no loss  (reference main.cpp:124-125,144-145)
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, double* s) {
    std::unique_ptr<CostFunction> own(cost);
    if (cost->kind != CostFunction::SWITCH_PRIOR || loss) throw std::invalid_argument("one parameter block: SwitchPriorResidue without loss only");
    prior_lambda_[s] = cost->dx;
  }
  void SetParameterBlockConstant(double* p) { fixed_ = block(p); }
  int NumResidualBlocks() const { return (int)ia_.size(); }
  int NumParameterBlocks() const { return (int)ptr_.size(); }

 private:
  friend struct SolverAccess;
  int32_t block(double* p) {
    auto it = id_.find(p);
    if (it != id_.end()) return it->second;
    int32_t k = (int32_t)ptr_.size();
    id_[p] = k;
    ptr_.push_back(p);
    return k;
  }
  std::unordered_map<double*, int32_t> id_;
  std::vector<double*> ptr_;
  std::vector<int32_t> ia_, ib_;
  std::vector<double> meas_;
  std::vector<uint8_t> kind_;
  int32_t fixed_ = -1;
  double delta_ = 0.0;
  std::vector<double*> switch_;                    // per residual block: its switch variable or nullptr
  std::unordered_map<double*, double> prior_lambda_;  // switch -> lambda of its prior
  bool any_dcs_ = false, any_sc_ = false, mixed_loss_ = false;
};

namespace Solver {
struct Options {
  bool minimizer_progress_to_stdout = false;
  LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;  // = the library's exact solve: the direct chain + low-rank solve where it applies
                                                                 // (INTEL, MIT, CSAIL, FR079 ...), else PCG to pcg_rtol; BLOCK_JACOBI_PCG: always PCG
  int max_num_iterations = 50;
  double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  double initial_trust_region_radius = 1e4;
  double pcg_rtol = 1e-10;
  int pcg_max_iters = 200000;
  int device = 0;
};
struct Summary {
  pgo_summary s{};
  std::vector<pgo_iter_record> iterations;
  int num_parameter_blocks = 0, num_residual_blocks = 0;
  std::string FullReport() const {
    static const char* term[] = {"?", "CONVERGENCE (function tolerance)", "CONVERGENCE (gradient tolerance)",
                                 "CONVERGENCE (parameter tolerance)", "NO_CONVERGENCE (max iterations)",
                                 "NO_CONVERGENCE (min trust region radius)", "FAILURE"};
    std::ostringstream o;
    char b[256];
    o << "\nSolver Summary (pgo-amd, MI355X HIP backend)\n\n";
    o << "Parameter blocks   " << num_parameter_blocks << "\nResidual blocks    " << num_residual_blocks << "\n";
    o << "Linear solver      block-Jacobi PCG (total " << s.total_pcg_iters << " iterations)\n\n";
    snprintf(b, sizeof b, "Cost:\nInitial  %.6e\nFinal    %.6e\nChange   %.6e\n\n", s.initial_cost, s.final_cost, s.initial_cost - s.final_cost);
    o << b;
    o << "Minimizer iterations  " << s.iterations << "\nSuccessful steps      " << s.successful_steps << "\n\n";
    snprintf(b, sizeof b, "Time (in seconds):\n  Residual+Jacobian  %.6f\n  Assembly           %.6f\n  Linear solver      %.6f\n  Candidate cost     %.6f\nTotal                %.6f\n\n",
             s.seconds_eval, s.seconds_assemble, s.seconds_linear, s.seconds_candidate, s.seconds_total);
    o << b << "Termination: " << term[(s.termination >= 0 && s.termination <= 6) ? s.termination : 0] << "\n";
    return o.str();
  }
};
}  // namespace Solver

struct SolverAccess {
  static void check(int st) {
    if (st != PGO_OK) throw std::runtime_error(std::string("pgo: ") + pgo_strerror(st) + ": " + pgo_last_error());
  }
  // Problem -> device handle (the arrays the residual blocks were recorded into)
  static pgo_t* Prepare(const Solver::Options& opt, Problem* pr) {
    if (pr->mixed_loss_) throw std::invalid_argument("this backend needs one shared loss for all residual blocks (as main.cpp:68)");
    const int32_t N = (int32_t)pr->ptr_.size(), E = (int32_t)pr->ia_.size();
    std::vector<double> poses((size_t)3 * N);
    for (int32_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) poses[3 * (size_t)i + k] = pr->ptr_[i][k];
    pgo_options o;
    pgo_options_default(&o);
    if (pr->any_sc_ && pr->any_dcs_) throw std::invalid_argument("DCS and switchable blocks cannot be mixed (the reference's METHOD is one of them)");
    o.method = pr->any_sc_ ? 2 : (pr->any_dcs_ ? 1 : 0);
    if (pr->any_sc_) {  // every switch needs exactly its prior, all with one lambda (as main.cpp:107)
      double lam = -1.0;
      for (double* sp : pr->switch_) {
        if (!sp) continue;
        auto it = pr->prior_lambda_.find(sp);
        if (it == pr->prior_lambda_.end()) throw std::invalid_argument("a switch variable has no SwitchPriorResidue");
        if (lam >= 0.0 && it->second != lam) throw std::invalid_argument("this backend needs one lambda for all switch priors");
        lam = it->second;
        if (*sp != 1.0) throw std::invalid_argument("switch variables must start at 1.0 (main.cpp:117,139)");
      }
      o.sc_prior_lambda = lam;
    }
    o.huber_delta = pr->delta_;
    o.fixed_pose = pr->fixed_;
    o.max_iters = opt.max_num_iterations;
    o.ftol = opt.function_tolerance;
    o.gtol = opt.gradient_tolerance;
    o.ptol = opt.parameter_tolerance;
    o.radius0 = opt.initial_trust_region_radius;
    o.pcg_rtol = opt.pcg_rtol;
    o.linear_solver = opt.linear_solver_type == BLOCK_JACOBI_PCG ? 1 : 0;
    o.pcg_max_iters = opt.pcg_max_iters;
    o.verbose = opt.minimizer_progress_to_stdout ? 1 : 0;
    pgo_t* h = nullptr;
    check(pgo_create(&h, N, poses.data(), E, pr->ia_.data(), pr->ib_.data(), pr->meas_.data(), pr->kind_.data(), &o, nullptr, opt.device));
    return h;
  }
  // results back into the caller's parameter blocks (in place, like Ceres); destroys the handle
  static void Finish(Problem* pr, pgo_t* h, int st, Solver::Summary* sum) {
    const int32_t N = (int32_t)pr->ptr_.size(), E = (int32_t)pr->ia_.size();
    std::vector<double> poses((size_t)3 * N);
    if (st == PGO_OK) st = pgo_get_poses(h, poses.data());
    std::vector<double> sw((size_t)E, 1.0);
    if (st == PGO_OK && pr->any_sc_) st = pgo_get_switches(h, sw.data(), nullptr);
    if (st == PGO_OK) {
      sum->iterations.resize((size_t)pgo_num_iter_records(h));
      st = pgo_get_iter_records(h, sum->iterations.data(), (int32_t)sum->iterations.size());
    }
    pgo_destroy(h);
    check(st);
    for (int32_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) pr->ptr_[i][k] = poses[3 * (size_t)i + k];
    for (int32_t e = 0; e < E; ++e)
      if (pr->switch_[e]) *pr->switch_[e] = sw[(size_t)e];
    sum->num_parameter_blocks = N;
    sum->num_residual_blocks = E;
  }
  static void Solve(const Solver::Options& opt, Problem* pr, Solver::Summary* sum) {
    pgo_t* h = Prepare(opt, pr);
    const int st = pgo_solve(h, &sum->s);
    Finish(pr, h, st, sum);
  }
  // The layer managers' pattern (src/simple_layer_manager.cpp:457-622: one ceres::Solve per candidate layer / window).
  // ONE batched handle (pgo_batch_*: a workgroup per problem, per-problem LM state on one launch sequence) when every
  // problem uses the same plain / DCS functor, loss and anchor index; else a pool of host threads over ordinary handles.
  static bool BatchEligible(const std::vector<Problem*>& prs) {
    if (prs.empty()) return false;
    for (Problem* pr : prs)
      if (pr->mixed_loss_ || pr->any_sc_ || pr->any_dcs_ != prs[0]->any_dcs_ || pr->delta_ != prs[0]->delta_ ||
          pr->fixed_ != prs[0]->fixed_ || pr->ptr_.empty())
        return false;
    return true;
  }
  static void SolveBatch(const Solver::Options& opt, const std::vector<Problem*>& prs, std::vector<Solver::Summary>* sums,
                         int max_concurrency) {
    if (BatchEligible(prs)) {
      std::vector<pgo_graph*> gs(prs.size(), nullptr);
      pgo_batch_t* b = nullptr;
      auto cleanup = [&] {
        for (pgo_graph* g : gs) pgo_graph_free(g);
        if (b) pgo_batch_destroy(b);
      };
      try {
        for (size_t i = 0; i < prs.size(); ++i) {
          Problem* pr = prs[i];
          const int32_t N = (int32_t)pr->ptr_.size(), E = (int32_t)pr->ia_.size();
          std::vector<double> poses((size_t)3 * N);
          for (int32_t k = 0; k < N; ++k)
            for (int c = 0; c < 3; ++c) poses[3 * (size_t)k + c] = pr->ptr_[k][c];
          check(pgo_graph_from_arrays(N, poses.data(), E, pr->ia_.data(), pr->ib_.data(), pr->meas_.data(), nullptr, pr->kind_.data(), &gs[i]));
        }
        pgo_options o;
        pgo_options_default(&o);
        o.method = prs[0]->any_dcs_ ? 1 : 0;
        o.huber_delta = prs[0]->delta_;
        o.fixed_pose = prs[0]->fixed_;
        o.max_iters = opt.max_num_iterations;
        o.ftol = opt.function_tolerance;
        o.gtol = opt.gradient_tolerance;
        o.ptol = opt.parameter_tolerance;
        o.radius0 = opt.initial_trust_region_radius;
        o.pcg_rtol = opt.pcg_rtol;
        o.linear_solver = opt.linear_solver_type == BLOCK_JACOBI_PCG ? 1 : 0;
        o.pcg_max_iters = opt.pcg_max_iters;
        int st = pgo_batch_create(&b, (int32_t)prs.size(), gs.data(), &o, opt.device);
        if (st != PGO_ERR_UNSUPPORTED) {
          check(st);
          std::vector<pgo_summary> raw(prs.size());
          check(pgo_batch_solve(b, raw.data()));
          sums->assign(prs.size(), Solver::Summary());
          for (size_t i = 0; i < prs.size(); ++i) {
            Problem* pr = prs[i];
            const int32_t N = (int32_t)pr->ptr_.size();
            std::vector<double> poses((size_t)3 * N);
            check(pgo_batch_get_poses(b, (int32_t)i, poses.data()));
            for (int32_t k = 0; k < N; ++k)
              for (int c = 0; c < 3; ++c) pr->ptr_[k][c] = poses[3 * (size_t)k + c];
            Solver::Summary& sm = (*sums)[i];
            sm.s = raw[i];
            sm.iterations.resize((size_t)pgo_batch_num_iter_records(b, (int32_t)i));
            check(pgo_batch_get_iter_records(b, (int32_t)i, sm.iterations.data(), (int32_t)sm.iterations.size()));
            sm.num_parameter_blocks = N;
            sm.num_residual_blocks = (int)pr->ia_.size();
          }
          cleanup();
          return;
        }
      } catch (...) {
        cleanup();
        throw;
      }
      cleanup();  // a problem the batched handle does not take (a row with > 256 edges): the thread pool below
    }
    std::vector<pgo_t*> hs;
    try {
      for (Problem* pr : prs) hs.push_back(Prepare(opt, pr));
    } catch (...) {
      for (pgo_t* h : hs) pgo_destroy(h);
      throw;
    }
    std::vector<pgo_summary> raw(prs.size());
    const int st = pgo_solve_batch(hs.data(), (int32_t)hs.size(), raw.data(), max_concurrency);
    const std::string msg = st != PGO_OK ? std::string(pgo_last_error()) : std::string();
    sums->assign(prs.size(), Solver::Summary());
    for (size_t i = 0; i < prs.size(); ++i) {
      (*sums)[i].s = raw[i];
      if (st == PGO_OK) Finish(prs[i], hs[i], PGO_OK, &(*sums)[i]);
      else pgo_destroy(hs[i]);
    }
    if (st != PGO_OK) throw std::runtime_error(std::string("pgo: ") + pgo_strerror(st) + ": " + msg);
  }
};

inline void Solve(const Solver::Options& opt, Problem* problem, Solver::Summary* summary) { SolverAccess::Solve(opt, problem, summary); }
// many independent problems at once; summaries->size() == problems.size() afterwards
inline void SolveBatch(const Solver::Options& opt, const std::vector<Problem*>& problems, std::vector<Solver::Summary>* summaries,
                       int max_concurrency = 8) {
  SolverAccess::SolveBatch(opt, problems, summaries, max_concurrency);
}

}  // namespace pgo

#endif
