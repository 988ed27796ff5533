#include "solver_handle.hip.h"

// ====================================================================== batched independent solves
// the reference's layer managers (METHOD 3/4) are control logic around thousands of small independent
// ceres::Solve calls -- a copy of the whole graph or a window per candidate layer / edge, plain functor + Huber, the
// first pose constant, 1-2 LM iterations each (src/simple_layer_manager.cpp:457-622, src/layer_manager.cpp:137-179,
// 602-654).  A pgo_batch is ONE handle over the block-diagonal union of n such problems: one launch of the fused edge
// kernel / the assembly kernel / the preconditioner set-up covers all of them, k_pcg_solo solves every problem's linear
// system in its own workgroup, and the TrustRegionMinimizer state (radius, cost, accept / reject, termination) is kept
// per problem.  About ten launches per LM iteration for the whole batch, whatever n is.
struct pgo_batch {
  std::unique_ptr<pgo_handle> U;
  int32_t n = 0;
  std::vector<int32_t> row0, npos, nedge;     // per problem: first row of the union, poses, edges
  struct State {
    bool active = true;
    int iter = 0, prev_success = 1, invalid_run = 0, successful = 0, total_pcg = 0, termination = 0;
    double cost = 0, initial_cost = 0, radius = 0, decrease_factor = 2, x_norm = 0, gmax = 0, seconds = 0;
    std::vector<pgo_iter_record> recs;
  };
  std::vector<State> st;
  std::vector<dev::SoloProb> h_prob;
  dev::SoloProb* d_prob = nullptr;
  dev::SoloOut* d_out = nullptr;
  dev::ProbRange* d_range = nullptr;
  dev::ProbSums* d_sums = nullptr;
  int32_t* d_accept = nullptr;
  std::vector<dev::SoloOut> h_out;
  std::vector<dev::ProbSums> h_sums;
  std::vector<int32_t> h_accept;
  std::vector<double> h_radius;
  bool begun = false;

  int reduce(bool with_cost, bool with_grad, const double* x) {
    pgo_handle& H = *U;
    hipLaunchKernelGGL(dev::k_prob_reduce<>, dim3(n), dim3(dev::WG), 0, H.stream, (const dev::ProbRange*)d_range,
                       with_cost ? (const double*)H.edge_cost : (const double*)nullptr,
                       with_grad ? (const double*)H.gs : (const double*)nullptr, (const double*)H.scale, x, H.S.lo, d_sums);
    PGOC(H.check_launch("k_prob_reduce"));
    HIPC(hipMemcpyAsync(h_sums.data(), d_sums, (size_t)n * sizeof(dev::ProbSums), hipMemcpyDeviceToHost, H.stream));
    return H.sync();
  }
  int begin();
  int iterate(bool* all_done);
};

int pgo_batch::begin() {
  pgo_handle& H = *U;
  HIPC(hipSetDevice(H.device));
  const pgo_options& o = H.opt;
  for (State& z : st) z = State();
  // iteration 0: unit scales -> column norms -> Jacobi scaling (per column, so per problem by construction)
  hipLaunchKernelGGL(dev::k_jacobi_scale<>, dim3(H.g_rows), dim3(dev::WG), 0, H.stream, H.hd, H.S.n_loc, H.S.lo, -1, 0, H.scale,
                     (const uint8_t*)H.fixed_mask);
  PGOC(H.check_launch("k_jacobi_scale"));
  PGOC(H.eval_enqueue(H.poses, nullptr, 1, true, 0));
  PGOC(H.assemble_enqueue());
  if (o.jacobi_scaling) {
    hipLaunchKernelGGL(dev::k_jacobi_scale<>, dim3(H.g_rows), dim3(dev::WG), 0, H.stream, H.hd, H.S.n_loc, H.S.lo, -1, 1, H.scale,
                       (const uint8_t*)H.fixed_mask);
    PGOC(H.check_launch("k_jacobi_scale"));
    PGOC(H.assemble_enqueue());
  }
  PGOC(reduce(true, true, H.poses));
  for (int k = 0; k < n; ++k) {
    State& z = st[k];
    z.cost = z.initial_cost = h_sums[k].cost;
    z.gmax = h_sums[k].gmax;
    z.x_norm = std::sqrt(h_sums[k].xnorm2);
    z.radius = o.radius0;
    pgo_iter_record R;
    memset(&R, 0, sizeof R);
    R.step_ok = 1;
    R.cost = z.cost;
    R.gradient_max_norm = z.gmax;
    R.radius = z.radius;
    z.recs.push_back(R);
    if (!std::isfinite(z.cost)) {  // "Residual and Jacobian evaluation failed" at the initial point
      z.termination = PGO_TERM_FAILURE;
      z.active = false;
    }
  }
  begun = true;
  return PGO_OK;
}

// one TrustRegionMinimizer iteration of every problem that is still running (same policy as pgo_handle::lm_iteration)
int pgo_batch::iterate(bool* all_done) {
  pgo_handle& H = *U;
  const pgo_options& o = H.opt;
  const double it0 = wall_s();
  int n_active = 0;
  for (int k = 0; k < n; ++k) {
    State& z = st[k];
    if (z.active) {
      if (z.iter >= o.max_iters) z.termination = PGO_TERM_NO_CONVERGENCE;
      else if (z.prev_success && z.gmax <= o.gtol) z.termination = PGO_TERM_CONVERGENCE_GTOL;
      else if (z.radius < o.min_radius) z.termination = PGO_TERM_MIN_RADIUS;
      if (z.termination) z.active = false;
    }
    h_prob[k].active = z.active ? 1 : 0;
    h_radius[k] = z.radius;
    n_active += z.active;
  }
  *all_done = n_active == 0;
  if (n_active == 0) return PGO_OK;
  HIPC(hipMemcpyAsync(d_prob, h_prob.data(), (size_t)n * sizeof(dev::SoloProb), hipMemcpyHostToDevice, H.stream));
  HIPC(hipMemcpyAsync(H.prob_radius, h_radius.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, H.stream));
  PGOC(H.prepare_system());
  dev::SoloArgs A;
  A.A = H.spmv_args(H.p_full, H.ap, H.part[0], 1, nullptr);
  A.V = H.cg_vec();
  A.C = H.chain_pre();
  if (!H.chain_len) A.C.cw = nullptr;
  A.chain_steps = H.solo_steps;
  A.scan_levels = H.solo_scan;
  A.b = H.gs;
  A.prob = d_prob;
  A.out = d_out;
  A.x = H.poses;
  A.scale = H.scale;
  A.cand = H.cand;
  hipLaunchKernelGGL(dev::k_pcg_solo<>, dim3(n), dim3(dev::SOLO_WG), 0, H.stream, A);
  PGOC(H.check_launch("k_pcg_solo"));
  HIPC(hipMemcpyAsync(h_out.data(), d_out, (size_t)n * sizeof(dev::SoloOut), hipMemcpyDeviceToHost, H.stream));
  // candidate cost of every problem (the rows of idle problems: cand was not written this iteration -- never read below)
  PGOC(H.eval_enqueue(H.cand, nullptr, 1, false, 0));
  PGOC(reduce(true, false, H.cand));
  bool any_accept = false;
  std::vector<pgo_iter_record> R((size_t)n);
  for (int k = 0; k < n; ++k) {
    h_accept[k] = 0;
    State& z = st[k];
    if (!z.active) continue;
    pgo_iter_record& r = R[k];
    memset(&r, 0, sizeof r);
    ++z.iter;
    r.iter = z.iter;
    const dev::SoloOut& q = h_out[k];
    z.total_pcg += q.iters;
    r.pcg_iters = q.iters;
    r.pcg_rel_residual = q.bb > 0.0 ? std::sqrt(q.rr / q.bb) : 0.0;
    const double model = q.ydotg - 0.5 * q.yHy;
    r.gradient_max_norm = z.gmax;
    if (!std::isfinite(model) || !std::isfinite(q.step2) || !(model > 0.0)) {  // invalid step
      if (++z.invalid_run >= 5) {
        z.termination = PGO_TERM_FAILURE;
        z.active = false;
        --z.iter;
        continue;
      }
      z.radius /= z.decrease_factor;
      z.decrease_factor *= 2.0;
      z.prev_success = 0;
      r.step_ok = -1;
      r.cost = z.cost;
      r.radius = z.radius;
      z.recs.push_back(r);
      continue;
    }
    z.invalid_run = 0;
    double cand_cost = h_sums[k].cost;
    if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
    r.step_norm = std::sqrt(q.step2);
    r.cost_change = z.cost - cand_cost;
    if (r.step_norm <= o.ptol * (z.x_norm + o.ptol) || std::fabs(r.cost_change) <= o.ftol * z.cost) {
      z.termination = (r.step_norm <= o.ptol * (z.x_norm + o.ptol)) ? PGO_TERM_CONVERGENCE_PTOL : PGO_TERM_CONVERGENCE_FTOL;
      z.active = false;
      r.cost = z.cost;
      r.radius = z.radius;
      z.recs.push_back(r);
      continue;
    }
    const double rho = (cand_cost >= std::numeric_limits<double>::max()) ? -std::numeric_limits<double>::max() : r.cost_change / model;
    r.relative_decrease = rho;
    if (rho > o.min_relative_decrease) {
      h_accept[k] = 1;
      any_accept = true;
      const double t = 2.0 * rho - 1.0;
      z.radius = std::min(o.max_radius, z.radius / std::max(1.0 / 3.0, 1.0 - t * t * t));
      z.decrease_factor = 2.0;
      z.prev_success = 1;
      ++z.successful;
      r.step_ok = 1;
    } else {
      z.radius /= z.decrease_factor;
      z.decrease_factor *= 2.0;
      z.prev_success = 0;
      r.step_ok = 0;
      r.cost = cand_cost;
      r.radius = z.radius;
      z.recs.push_back(r);
    }
  }
  if (any_accept) {
    HIPC(hipMemcpyAsync(d_accept, h_accept.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, H.stream));
    hipLaunchKernelGGL(dev::k_accept_rows<>, dim3(H.g_flat), dim3(dev::WG), 0, H.stream, H.S.n_loc, H.S.lo, (const int32_t*)H.prob_of_256,
                       (const int32_t*)d_accept, (const double*)H.cand, H.poses);
    PGOC(H.check_launch("k_accept_rows"));
    // re-linearise everything: the problems that did not move reproduce their records and blocks bit for bit
    PGOC(H.eval_enqueue(H.poses, nullptr, 1, true, 0));
    PGOC(H.assemble_enqueue());
    PGOC(reduce(true, true, H.poses));
    for (int k = 0; k < n; ++k) {
      if (!h_accept[k]) continue;
      State& z = st[k];
      pgo_iter_record& r = R[k];
      if (!std::isfinite(h_sums[k].cost)) {  // non-finite Jacobian at an accepted point (the asin' singularity)
        z.termination = PGO_TERM_FAILURE;
        z.active = false;
      } else {
        z.cost = h_sums[k].cost;
        z.gmax = h_sums[k].gmax;
        z.x_norm = std::sqrt(h_sums[k].xnorm2);
      }
      r.cost = z.cost;
      r.gradient_max_norm = z.gmax;
      r.radius = z.radius;
      z.recs.push_back(r);
    }
  }
  const double dt = wall_s() - it0;
  for (int k = 0; k < n; ++k)
    if (h_prob[k].active) {
      st[k].seconds += dt;
      if (!st[k].recs.empty()) st[k].recs.back().seconds = dt;
    }
  return PGO_OK;
}

extern "C" {

int pgo_batch_create(pgo_batch_t** out, int32_t n, const pgo_graph* const* graphs, const pgo_options* opt, int device) {
  if (!out || n <= 0 || !graphs) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_create: bad argument");
  pgo_options o;
  if (opt) o = *opt;
  else pgo_options_default(&o);
  if (o.method != 0 && o.method != 1) return fail(PGO_ERR_UNSUPPORTED, "pgo_batch: METHOD 0 and 1 only");
  if (o.info_weighting) return fail(PGO_ERR_UNSUPPORTED, "pgo_batch: info_weighting is not supported");
  if (o.pcg_block_poses > 1) return fail(PGO_ERR_UNSUPPORTED, "pgo_batch: the chain or the 3x3 block-Jacobi preconditioner only");
  PGOC(require_device(device));
  std::unique_ptr<pgo_batch> B(new pgo_batch);
  B->n = n;
  B->row0.resize(n);
  B->npos.resize(n);
  B->nedge.resize(n);
  int64_t rows = 0, edges = 0;
  int32_t big = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (!graphs[k] || graphs[k]->g.n_poses() <= 0) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_create: problem " + std::to_string(k) + " is empty");
    const pgo::Graph& G = graphs[k]->g;
    if (o.fixed_pose >= G.n_poses()) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_create: fixed_pose out of range in problem " + std::to_string(k));
    B->row0[k] = (int32_t)rows;
    B->npos[k] = G.n_poses();
    B->nedge[k] = G.n_edges();
    rows += ((int64_t)G.n_poses() + 255) / 256 * 256;   // every problem starts on a 256-row boundary
    edges += G.n_edges();
    if (G.n_poses() > graphs[big]->g.n_poses()) big = k;
    if (rows > (int64_t)1 << 30 || edges > (int64_t)1 << 30) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_create: batch too large");
  }
  // the union: poses (padding rows at the origin, constant), edges shifted to the union's numbering
  std::vector<double> poses((size_t)3 * rows, 0.0), meas((size_t)3 * edges);
  std::vector<int32_t> ia((size_t)edges), ib((size_t)edges);
  std::vector<uint8_t> kind((size_t)edges);
  std::unique_ptr<pgo_handle> H(new pgo_handle);
  H->fixed_mask_h.assign((size_t)rows, 1);
  int64_t eo = 0;
  for (int32_t k = 0; k < n; ++k) {
    const pgo::Graph& G = graphs[k]->g;
    const int32_t r0 = B->row0[k];
    memcpy(&poses[(size_t)3 * r0], G.pose.data(), (size_t)3 * G.n_poses() * sizeof(double));
    for (int32_t i = 0; i < G.n_poses(); ++i) H->fixed_mask_h[(size_t)r0 + i] = (i == o.fixed_pose) ? 1 : 0;
    for (int32_t e = 0; e < G.n_edges(); ++e) {
      if (G.ea[e] < 0 || G.ea[e] >= G.n_poses() || G.eb[e] < 0 || G.eb[e] >= G.n_poses() || G.ea[e] == G.eb[e])
        return fail(PGO_ERR_INVALID_ARG, "pgo_batch_create: problem " + std::to_string(k) + ", edge " + std::to_string(e) + ": bad endpoints");
      ia[(size_t)eo + e] = r0 + G.ea[e];
      ib[(size_t)eo + e] = r0 + G.eb[e];
      kind[(size_t)eo + e] = G.kind[e];
    }
    if (G.n_edges()) memcpy(&meas[(size_t)3 * eo], G.meas.data(), (size_t)3 * G.n_edges() * sizeof(double));
    if (k > 0) H->tile_breaks_h.push_back(r0);
    eo += G.n_edges();
  }
  // one preconditioner for the whole batch: what the library would choose for the largest problem alone (the dense
  // pose-block form has no one-workgroup kernel: 64-pose chain segments stand in for it)
  const pgo::Graph& GB = graphs[big]->g;
  int chain = pgo::resolve_chain_len(o.pcg_chain_len, o.pcg_block_poses, GB.n_poses(), GB.n_edges(), GB.ea.data(), GB.eb.data());
  if (chain == 0 && o.pcg_block_poses != 1) chain = 64;
  o.pcg_chain_len = chain;
  o.pcg_block_poses = 1;
  o.fixed_pose = -1;       // the mask carries one anchor per problem
  o.pose_ordering = 0;
  H->opt = o;
  H->comm = nullptr;
  H->device = device;
  H->batch_mode = true;
  PGOC(H->create((int32_t)rows, poses.data(), (int32_t)edges, ia.data(), ib.data(), meas.data(), nullptr, kind.data()));
  // per-problem ranges in the handle's local edge order (sorted by smaller endpoint => contiguous per problem) and tiles
  std::vector<dev::ProbRange> rng((size_t)n);
  std::vector<int32_t> p256((size_t)(rows / 256));
  B->h_prob.resize(n);
  {
    const pgo::ShardStructure& S = H->S;
    int32_t e = 0, t = 0;
    for (int32_t k = 0; k < n; ++k) {
      const int32_t r0 = B->row0[k], r1 = (k + 1 < n) ? B->row0[k + 1] : (int32_t)rows;
      rng[k].row0 = r0;
      rng[k].nrows = B->npos[k];
      rng[k].e0 = e;
      while (e < S.n_edges_local && std::min(S.ia[e], S.ib[e]) < r1) ++e;
      rng[k].e1 = e;
      while (t < S.n_tiles() && S.tile_row[t] < r0) ++t;
      if (t >= S.n_tiles() || S.tile_row[t] != r0) return fail(PGO_ERR_HIP, "pgo_batch_create: internal: tile boundaries");
      const int32_t t0 = t;
      while (t < S.n_tiles() && S.tile_row[t] < r1) ++t;
      dev::SoloProb& P = B->h_prob[k];
      P.row0 = r0;
      P.nrows = B->npos[k];
      P.tile0 = t0;
      P.ntiles = t - t0;
      P.active = 1;
      P.max_it = std::max(0, o.pcg_max_iters);
      P.rtol = o.pcg_rtol;
      for (int32_t b = r0 / 256; b < r1 / 256; ++b) p256[b] = k;
    }
  }
  PGOC(H->dalloc(&H->prob_of_256, (int64_t)p256.size()));
  PGOC(H->upload(H->prob_of_256, p256));
  PGOC(H->dalloc(&H->prob_radius, n));
  PGOC(H->dalloc(&B->d_prob, n));
  PGOC(H->dalloc(&B->d_out, n));
  PGOC(H->dalloc(&B->d_range, n));
  PGOC(H->dalloc(&B->d_sums, n));
  PGOC(H->dalloc(&B->d_accept, n));
  PGOC(H->upload(B->d_range, rng));
  PGOC(H->sync());  // rng / p256 die with this scope
  B->h_out.resize(n);
  B->h_sums.resize(n);
  B->h_accept.assign(n, 0);
  B->h_radius.assign(n, 0.0);
  B->st.resize(n);
  B->U = std::move(H);
  *out = B.release();
  return PGO_OK;
}

void pgo_batch_destroy(pgo_batch_t* b) { delete b; }

int pgo_batch_solve(pgo_batch_t* b, pgo_summary* summaries) {
  if (!b) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_solve: null");
  PGOC(b->begin());
  bool done = false;
  while (!done) PGOC(b->iterate(&done));
  if (summaries)
    for (int32_t k = 0; k < b->n; ++k) {
      const pgo_batch::State& z = b->st[k];
      pgo_summary& s = summaries[k];
      memset(&s, 0, sizeof s);
      s.termination = z.termination;
      s.iterations = z.iter;
      s.successful_steps = z.successful;
      s.total_pcg_iters = z.total_pcg;
      s.initial_cost = z.initial_cost;
      s.final_cost = z.cost;
      s.seconds_total = z.seconds;
    }
  return PGO_OK;
}

int32_t pgo_batch_size(const pgo_batch_t* b) { return b ? b->n : 0; }

int pgo_batch_get_poses(pgo_batch_t* b, int32_t k, double* out) {
  if (!b || !out || k < 0 || k >= b->n) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_get_poses: bad argument");
  pgo_handle& H = *b->U;
  HIPC(hipSetDevice(H.device));
  HIPC(hipMemcpyAsync(out, H.poses + 3 * (int64_t)b->row0[k], (size_t)3 * b->npos[k] * sizeof(double), hipMemcpyDeviceToHost, H.stream));
  return H.sync();
}

int pgo_batch_set_poses(pgo_batch_t* b, int32_t k, const double* poses) {
  if (!b || !poses || k < 0 || k >= b->n) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_set_poses: bad argument");
  pgo_handle& H = *b->U;
  HIPC(hipSetDevice(H.device));
  HIPC(hipMemcpyAsync(H.poses + 3 * (int64_t)b->row0[k], poses, (size_t)3 * b->npos[k] * sizeof(double), hipMemcpyHostToDevice, H.stream));
  b->begun = false;
  return H.sync();
}

int32_t pgo_batch_num_iter_records(const pgo_batch_t* b, int32_t k) {
  return (b && k >= 0 && k < b->n) ? (int32_t)b->st[k].recs.size() : 0;
}
int pgo_batch_get_iter_records(const pgo_batch_t* b, int32_t k, pgo_iter_record* out, int32_t cap) {
  if (!b || !out || k < 0 || k >= b->n) return fail(PGO_ERR_INVALID_ARG, "pgo_batch_get_iter_records: bad argument");
  const int32_t m = std::min<int32_t>(cap, (int32_t)b->st[k].recs.size());
  memcpy(out, b->st[k].recs.data(), (size_t)m * sizeof(pgo_iter_record));
  return PGO_OK;
}

}  // extern "C"

