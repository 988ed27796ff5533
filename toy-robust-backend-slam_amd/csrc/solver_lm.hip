// The minimiser: Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy defaults, i.e. what ceres::Solve does
// for DCS-ceres/main.cpp:154-163 -- Jacobi scaling, LM diagonal, step, model decrease, accept / reject, radius update,
// termination tests -- around the linear solve (solver_pcg.hip / solver_direct.hip).
#include "solver_handle.hip.h"

// evaluate (K1, unscaled records) + assemble (K2, applies the current `scale`) at `poses`;
// leaves cost/bad in h_scal[0..1]
int pgo_handle::linearize(bool reuse_records, bool assemble) {
  double t0 = wall_s();
  if (!reuse_records) {
    PGOC(eval_enqueue(poses, sw, 1, true, 0));
    PGOC(fetch_scal(0, 2));
    t_eval += wall_s() - t0;
    if (h_scal[1] > 0.0 || !std::isfinite(h_scal[0])) return fail(PGO_ERR_NUMERIC, "residual/Jacobian evaluation produced non-finite values");
  }
  if (!assemble) return PGO_OK;
  t0 = wall_s();
  PGOC(assemble_enqueue());
  PGOC(sync());
  t_asm += wall_s() - t0;
  return PGO_OK;
}

// METHOD 2, at the top of every LM iteration (the radius has changed): elimination coefficients of the switches for
// the current radius, re-assembly of the reduced pose system, gradient max-norm over poses AND switches, sum s^2.
int pgo_handle::refresh_switch_system() {
  const int g_sw = std::min(std::max(1, (S.n_edges_local + dev::WG - 1) / dev::WG), 1024);
  hipLaunchKernelGGL(dev::k_switch_prepare<>, dim3(g_sw), dim3(dev::WG), 0, stream, switch_arrays(), (const double*)jr, radius,
                     opt.min_lm_diagonal, opt.max_lm_diagonal, part[2], part[3]);
  PGOC(check_launch("k_switch_prepare"));
  PGOC(assemble_enqueue());
  hipLaunchKernelGGL(dev::k_grad_max<>, dim3(g_flat), dim3(dev::WG), 0, stream, (const double*)gs_full, (const double*)scale, S.n_loc,
                     S.lo, part[0]);
  PGOC(check_launch("k_grad_max"));
  PGOC(reduce_to_scal({{part[0], g_flat, 1}, {part[2], g_sw, 1}}, 10, true));
  PGOC(reduce_to_scal({{part[3], g_sw, 0}}, 12));
  PGOC(fetch_scal(10, 3));
  gmax = std::max(h_scal[10], h_scal[11]);
  sw_norm2 = h_scal[12];
  x_norm = std::sqrt(xnorm2_pose + sw_norm2);
  sw_fresh = true;
  return PGO_OK;
}

int pgo_handle::lm_begin() {
  HIPC(hipSetDevice(device));
  lm_active = true;
  lm_done = false;
  iter = 0;
  prev_success = 1;
  invalid_run = 0;
  successful = 0;
  total_pcg = 0;
  termination = 0;
  radius = opt.radius0;
  decrease_factor = 2.0;
  last_pcg_iters = 0;
  if (dl_possible) {   // every solve of the handle takes the same solver decisions (lm_iteration)
    direct = false;
    dl_last_probe = dl_dear_run = 0;
  }
  t_eval = t_asm = t_lin = t_cand = 0;
  recs.clear();
  const double t_begin = wall_s();
  t_total = 0;
  const int fixed = fixed_internal;
  if (has_sw) {  // switches start at 1.0 (main.cpp:117,139); nothing eliminated yet
    std::vector<double> ones((size_t)std::max(1, S.n_edges_local), 1.0);
    HIPC(hipMemcpyAsync(sw, ones.data(), (size_t)S.n_edges_local * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPC(hipMemcpyAsync(sw_cand, ones.data(), (size_t)S.n_edges_local * sizeof(double), hipMemcpyHostToDevice, stream));
    HIPC(hipMemsetAsync(sw_c, 0, (size_t)S.n_edges_local * sizeof(double), stream));
    HIPC(hipMemsetAsync(sw_gamma, 0, (size_t)S.n_edges_local * sizeof(double), stream));
    PGOC(sync());  // `ones` dies with this scope
  }
  // pass 1: unit scales (0 on the constant pose) -> column norms for Jacobi scaling
  hipLaunchKernelGGL(dev::k_jacobi_scale<>, dim3(g_rows), dim3(dev::WG), 0, stream, hd, S.n_loc, S.lo, fixed, 0, scale, (const uint8_t*)fixed_mask);
  PGOC(check_launch("k_jacobi_scale"));
  PGOC(allgather(scale));
  PGOC(linearize(false));
  const double cost0 = h_scal[0];
  if (has_sw) {  // Jacobi scale of the switch columns from the iteration-0 Jacobian
    hipLaunchKernelGGL(dev::k_switch_scale<>, dim3((S.n_edges_local + 255) / 256 + 1), dim3(256), 0, stream, switch_arrays(),
                       opt.jacobi_scaling);
    PGOC(check_launch("k_switch_scale"));
  }
  if (opt.jacobi_scaling) {
    hipLaunchKernelGGL(dev::k_jacobi_scale<>, dim3(g_rows), dim3(dev::WG), 0, stream, hd, S.n_loc, S.lo, fixed, 1, scale, (const uint8_t*)fixed_mask);
    PGOC(check_launch("k_jacobi_scale"));
    PGOC(allgather(scale));
    PGOC(linearize(true));  // the records do not depend on the scales: re-assemble only
  }
  cost = initial_cost = cost0;
  hipLaunchKernelGGL(dev::k_grad_max<>, dim3(g_flat), dim3(dev::WG), 0, stream, gs, scale, S.n_loc, S.lo, part[0]);
  PGOC(check_launch("k_grad_max"));
  PGOC(reduce_to_scal({{part[0], g_flat, 1}}, 2, true));
  hipLaunchKernelGGL(dev::k_xnorm<>, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, poses, scale, part[1]);
  PGOC(check_launch("k_xnorm"));
  PGOC(reduce_to_scal({{part[1], g_flat, 0}}, 3));
  PGOC(fetch_scal(2, 2));
  gmax = h_scal[2];
  xnorm2_pose = h_scal[3];
  x_norm = std::sqrt(xnorm2_pose);
  sw_fresh = false;
  if (has_sw) PGOC(refresh_switch_system());
  lin_valid = true;
  pgo_iter_record R;
  memset(&R, 0, sizeof R);
  R.iter = 0;
  R.step_ok = 1;
  R.cost = cost;
  R.gradient_max_norm = gmax;
  R.radius = radius;
  R.seconds = wall_s() - t_begin;
  t_total += R.seconds;
  recs.push_back(R);
  if (opt.verbose) {
    printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  pcg_it  pcg_rel\n");
    printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e  %6d  %.1e\n", 0, cost, 0.0, gmax, 0.0, 0.0, radius, 0, 0.0);
  }
  return PGO_OK;
}

// one TrustRegionMinimizer iteration.  *stop is set when a termination test fires.
int pgo_handle::lm_iteration(bool* stop) {
  *stop = false;
  // FinalizeIterationAndCheckIfMinimizerCanContinue
  if (iter >= opt.max_iters) {
    termination = PGO_TERM_NO_CONVERGENCE;
    *stop = true;
    return PGO_OK;
  }
  if (has_sw && !sw_fresh) PGOC(refresh_switch_system());  // the radius changed since the last assembly (rejected / invalid step)
  if (prev_success && gmax <= opt.gtol) {
    termination = PGO_TERM_CONVERGENCE_GTOL;
    *stop = true;
    return PGO_OK;
  }
  if (radius < opt.min_radius) {
    termination = PGO_TERM_MIN_RADIUS;
    *stop = true;
    return PGO_OK;
  }
  const double it0 = wall_s();
  ++iter;
  pgo_iter_record R;
  memset(&R, 0, sizeof R);
  R.iter = iter;

  // LM diagonal + preconditioner, then the linear solve
  double t0 = wall_s();
  PGOC(prepare_system());
  int k_it = 0;
  double rel = 0.0;
  // Ranks above DIRECT_AUTO_RANK in auto mode (dl_possible): which solver is cheaper depends on the conditioning and changes
  // along the trajectory (M3500 without DCS: ~1000 PCG iterations per LM iteration at first, < 100 later; with DCS 1200-2300
  // throughout), so the handle decides from what it sees -- from iteration COUNTS, not clocks: reproducible --
  //   on PCG:    two consecutive solves dearer than a direct solve of this rank  -> the direct solve takes over;
  //   on direct: every DIRECT_PROBE_EVERY-th LM iteration is solved by PCG; if that was cheaper, PCG takes over again.
  const bool probe = direct && dl_possible && iter - dl_last_probe >= DIRECT_PROBE_EVERY;
  const bool run_direct = direct && !probe;
  if (run_direct) {
    PGOC(direct_solve());
  } else {
    if (direct) PGOC(prepare_preconditioner());
    PGOC(pcg(&k_it, &rel));
  }
  if (probe) dl_retry = true;   // (this iteration's step is PCG's: no direct-solve residual, no fallback)
  const int st_tail = lm_iteration_tail(stop, R, it0, t0, k_it, rel);
  if (probe) dl_retry = false;
  if (st_tail == PGO_OK && dl_possible && !run_direct && !*stop) {
    // a PCG iteration: 14 us on graphs of a few thousand poses (two launches), 39 us at 100k poses
    const bool dear = (double)k_it * (PCG_SECONDS_PER_ITER_SMALL + 0.25e-9 * S.n_poses) > dl_est_seconds;
    if (probe) {
      dl_last_probe = iter;
      if (!dear) {
        direct = false;
        dl_dear_run = 0;
        if (opt.verbose) printf("pgo: %d PCG iterations in LM iteration %d: back to PCG\n", k_it, iter);
      }
    } else {
      dl_dear_run = dear ? dl_dear_run + 1 : 0;
      if (dl_dear_run >= 2) {
        // The direct solver's buffers (~1 GB at rank 5862, more with long chains) are allocated HERE, in the middle of a
        // solve that PCG is handling: a failed allocation must not turn a speed-up heuristic into a failed pgo_solve.
        // Any failure -> the partial buffers are freed, the handle stays on PCG for good, the error text is cleared.
        bool ok = true;
        if (!dl_ready) {
          const size_t mark = allocs.size();
          const int64_t bytes_mark = device_bytes;
          if (direct_setup(S.n_poses, true) != PGO_OK || !dl_ready) {
            (void)hipStreamSynchronize(stream);   // uploads into the buffers about to be freed
            (void)hipGetLastError();
            for (size_t k = mark; k < allocs.size(); ++k) (void)hipFree(allocs[k]);
            allocs.resize(mark);
            device_bytes = bytes_mark;
            clear_direct_buffers();
            dl_possible = false;
            direct = false;
            ok = false;
            (void)fail(PGO_OK, "");
            if (opt.verbose) printf("pgo: the direct solver could not be set up (LM iteration %d): staying on PCG\n", iter);
          }
        }
        if (ok) {
          direct = true;
          dl_last_probe = iter;
          if (dl_switched_at == 0) dl_switched_at = iter;
          if (opt.verbose) printf("pgo: %d PCG iterations in LM iteration %d: the direct solve takes over (rank %d)\n", k_it, iter, dl_K);
        }
      }
    }
  }
  return st_tail;
}

int pgo_handle::lm_iteration_tail(bool* stop, pgo_iter_record& R, double it0, double t0, int k_it, double rel) {
  total_pcg += k_it;
  R.pcg_iters = k_it;
  R.pcg_rel_residual = rel;
  // model_cost_change = -(J d).(r + J d / 2), d = -S y   ==   y.gs - y.(H y) / 2
  if (!solo) {  // (the one-workgroup solve has done all of this in its epilogue; the gather vector holds y either way)
    // The model decrease below uses r = b - (H + D'D) y.  PCG's recurrence residual is that up to rounding drift, which
    // grows with the iteration count: in the exact mode (tight tolerance, up to 1e5 iterations on the ill-conditioned
    // late systems) the drift would bias rho and with it the accept / reject and radius decisions, unnoticed -- so
    // there the residual is recomputed with one product (nothing next to the solve it follows).  The inexact mode
    // (rtol 0.1, ~100 iterations) keeps the recurrence residual; the direct solve writes the true residual itself.
    const bool true_residual = k_it > 0 && (opt.pcg_rtol < 1e-6 || k_it > 1000);
    if (has_sw || true_residual) {  // (the switch back-substitution below reads y of both endpoints from the gather vector)
      hipLaunchKernelGGL(dev::k_scatter_owned<>, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, y, p_full);
      PGOC(check_launch("k_scatter_owned"));
      PGOC(share_gather_vector(p_full));
    }
    if (true_residual) {
      PGOC(spmv_enqueue(p_full, ap, part[0], 1, nullptr));
      hipLaunchKernelGGL(dev::k_dlr_resid<>, dim3(std::max<int64_t>(1, (3 * S.n_loc + 255) / 256)), dim3(256), 0, stream, (int64_t)3 * S.n_loc,
                         (const double*)gs, (const double*)ap, r);   // (at least one workgroup: a rank may own no rows)
      PGOC(check_launch("k_dlr_resid"));
    }
    // y.(H y) = y.b - y.r - y.(D y) from the residual (no further product by H): part[1] = y.b, part[0] = y.r, part[5] = y.(D y)
    hipLaunchKernelGGL(dev::k_model_terms<>, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * S.n_loc, (const double*)y, (const double*)gs,
                       (const double*)r, (const double*)d2, part[1], part[0], part[5]);
    PGOC(check_launch("k_model_terms"));
    // candidate x + d and |d|^2
    double* x_old = poses;
    hipLaunchKernelGGL(dev::k_candidate<>, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, x_old, scale, y, cand, part[3]);
    PGOC(check_launch("k_candidate"));
  }
  double model_sw = 0.0, step2_sw = 0.0;
  if (has_sw) {  // back-substitute the switches (needs y of both endpoints: in the gather vector after the share above)
    const int g_sw = std::min(std::max(1, (S.n_edges_local + dev::WG - 1) / dev::WG), 1024);
    hipLaunchKernelGGL(dev::k_switch_backsub<>, dim3(g_sw), dim3(dev::WG), 0, stream, switch_arrays(), (const int32_t*)e_ia,
                       (const int32_t*)e_ib, (const double*)jr, (const double*)scale, (const double*)p_full, part[2], part[4]);
    PGOC(check_launch("k_switch_backsub"));
    PGOC(reduce_to_scal({{part[2], g_sw, 0}, {part[4], g_sw, 0}}, 13));
  }
  if (solo) {
    h_scal[0] = h_solo->ydotg;
    h_scal[1] = h_solo->yHy;
    h_scal[2] = h_solo->step2;
  } else {
    PGOC(reduce_to_scal({{part[1], g_flat, 0}, {part[0], g_flat, 0}, {part[3], g_flat, 0}, {part[5], g_flat, 0}}, 0));
    // the candidate's cost is evaluated in the same breath (scal[6..7]; wasted only when the step turns out invalid): one
    // host synchronisation for the model terms AND the candidate instead of two
    PGOC(allgather(cand));
    PGOC(eval_enqueue(cand, sw_cand, 1, false, 6));
    PGOC(fetch_scal(0, has_sw ? 15 : 10));
    h_scal[1] = h_scal[0] - h_scal[1] - h_scal[3];   // y.(H y)
    if (direct && !dl_retry) R.pcg_rel_residual = dl_rel = (h_scal[9] > 0.0) ? std::sqrt(h_scal[8] / h_scal[9]) : 0.0;
  }
  if (has_sw) {
    if (solo) PGOC(fetch_scal(13, 2));
    model_sw = h_scal[13];
    step2_sw = h_scal[14];
  }
  t_lin += wall_s() - t0;
  const double ydotg = h_scal[0], yHy = h_scal[1], step2 = h_scal[2] + step2_sw;
  const double model = ydotg - 0.5 * yHy + model_sw;
  if (!std::isfinite(model) || !std::isfinite(step2) || !(model > 0.0)) {  // invalid step
    if (direct && !dl_retry) {
      // the direct solve produced no usable step (a capacitance matrix that lost positive definiteness to rounding, a
      // residual the refinement could not repair): this LM iteration is redone by PCG before Ceres' invalid-step rule applies
      dl_retry = true;
      ++dl_fallbacks;
      int k2 = 0;
      double rel2 = 0.0;
      int st2 = prepare_preconditioner();
      if (st2 == PGO_OK) st2 = pcg(&k2, &rel2);
      if (st2 == PGO_OK) st2 = lm_iteration_tail(stop, R, it0, t0, k2, rel2);
      dl_retry = false;
      return st2;
    }
    if (++invalid_run >= 5) {
      termination = PGO_TERM_FAILURE;
      *stop = true;
      return PGO_OK;
    }
    radius /= decrease_factor;
    decrease_factor *= 2.0;
    sw_fresh = false;
    prev_success = 0;
    R.step_ok = -1;
    R.cost = cost;
    R.radius = radius;
    R.gradient_max_norm = gmax;
    R.seconds = wall_s() - it0;
    t_total += R.seconds;
    recs.push_back(R);
    return PGO_OK;
  }
  invalid_run = 0;
  t0 = wall_s();
  if (solo) {   // (the one-workgroup solve synchronised inside pcg(): its candidate is evaluated here)
    PGOC(allgather(cand));
    PGOC(eval_enqueue(cand, sw_cand, 1, false, 6));
    PGOC(fetch_scal(6, 2));
  }
  t_cand += wall_s() - t0;
  double cand_cost = h_scal[6];
  if (h_scal[7] > 0.0 || !std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
  R.step_norm = std::sqrt(step2);
  R.cost_change = cost - cand_cost;
  R.gradient_max_norm = gmax;
  auto finish = [&](int term) {
    termination = term;
    R.cost = cost;
    R.radius = radius;
    R.seconds = wall_s() - it0;
    t_total += R.seconds;
    recs.push_back(R);
    *stop = true;
  };
  if (R.step_norm <= opt.ptol * (x_norm + opt.ptol)) {  // ParameterToleranceReached
    finish(PGO_TERM_CONVERGENCE_PTOL);
    return PGO_OK;
  }
  if (std::fabs(R.cost_change) <= opt.ftol * cost) {  // FunctionToleranceReached
    finish(PGO_TERM_CONVERGENCE_FTOL);
    return PGO_OK;
  }
  const double rho = (cand_cost >= std::numeric_limits<double>::max()) ? -std::numeric_limits<double>::max() : R.cost_change / model;
  R.relative_decrease = rho;
  if (rho > opt.min_relative_decrease) {  // HandleSuccessfulStep
    std::swap(poses, cand);
    if (has_sw) std::swap(sw, sw_cand);
    hipLaunchKernelGGL(dev::k_xnorm<>, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, poses, scale, part[1]);
    PGOC(check_launch("k_xnorm"));
    PGOC(reduce_to_scal({{part[1], g_flat, 0}}, 3));
    const double t = 2.0 * rho - 1.0;
    if (has_sw) {
      int st_lin = linearize(false, false);  // METHOD 2 assembles in refresh_switch_system(), with the new radius
      if (st_lin == PGO_ERR_NUMERIC) {
        finish(PGO_TERM_FAILURE);
        return PGO_OK;
      }
      PGOC(st_lin);
      cost = h_scal[0];
      radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
      radius = std::min(opt.max_radius, radius);
      PGOC(fetch_scal(3, 1));
      xnorm2_pose = h_scal[3];
      PGOC(refresh_switch_system());  // gmax over poses and switches, x_norm, reduced system for the next iteration
    } else {
      // K1, K2 and the gradient norm are enqueued together and fetched with ONE host synchronisation (three before: a
      // synchronisation is ~25 us of idle GPU, 4 % of an LM iteration on INTEL); if K1 reports a non-finite value the
      // assembled system is discarded with the step, as before
      const double tl0 = wall_s();
      PGOC(eval_enqueue(poses, sw, 1, true, 0));
      PGOC(assemble_enqueue());
      hipLaunchKernelGGL(dev::k_grad_max<>, dim3(g_flat), dim3(dev::WG), 0, stream, gs, scale, S.n_loc, S.lo, part[0]);
      PGOC(check_launch("k_grad_max"));
      PGOC(reduce_to_scal({{part[0], g_flat, 1}}, 2, true));
      PGOC(fetch_scal(0, 4));
      t_eval += wall_s() - tl0;  // (K2 and the norms included: no host synchronisation separates them any more)
      if (h_scal[1] > 0.0 || !std::isfinite(h_scal[0])) {
        (void)fail(PGO_ERR_NUMERIC, "residual/Jacobian evaluation produced non-finite values");
        finish(PGO_TERM_FAILURE);
        return PGO_OK;
      }
      cost = h_scal[0];
      radius = radius / std::max(1.0 / 3.0, 1.0 - t * t * t);
      radius = std::min(opt.max_radius, radius);
      gmax = h_scal[2];
      xnorm2_pose = h_scal[3];
      x_norm = std::sqrt(xnorm2_pose);
    }
    decrease_factor = 2.0;
    prev_success = 1;
    ++successful;
    R.step_ok = 1;
    R.cost = cost;
    R.gradient_max_norm = gmax;
  } else {  // HandleUnsuccessfulStep
    radius /= decrease_factor;
    decrease_factor *= 2.0;
    sw_fresh = false;
    prev_success = 0;
    R.step_ok = 0;
    R.cost = cand_cost;
  }
  R.radius = radius;
  R.seconds = wall_s() - it0;
  t_total += R.seconds;
  recs.push_back(R);
  if (opt.verbose)
    printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e  %6d  %.1e\n", iter, R.cost, R.cost_change, gmax, R.step_norm, rho,
           radius, k_it, rel);
  return PGO_OK;
}

void pgo_handle::fill_summary(pgo_summary* s) const {
  if (!s) return;
  memset(s, 0, sizeof *s);
  s->termination = termination;
  s->iterations = iter;
  s->successful_steps = successful;
  s->total_pcg_iters = total_pcg;
  s->initial_cost = initial_cost;
  s->final_cost = cost;
  s->seconds_total = t_total;
  s->seconds_eval = t_eval;
  s->seconds_assemble = t_asm;
  s->seconds_linear = t_lin;
  s->seconds_candidate = t_cand;
}

