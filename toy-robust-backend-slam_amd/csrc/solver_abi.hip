// The [gpu] part of the C-ABI (include/pgo.h): handle life cycle, evaluation, solves, results, test hooks, and the debug /
// kernel-bench entry points the parity tests and bench.py use.
#include "solver_handle.hip.h"

// ====================================================================== C-ABI
namespace {
struct Knob {
  const char* name;
  std::atomic<long long> value;
};
Knob g_knobs[] = {{"spmv_pipe", {-1}}, {"fused_p", {-1}}, {"direct_fail_at", {-1}}, {"direct_setup_fail", {-1}}, {"single_reduction", {-1}}, {"verify_residual", {-1}}, {"shm_timeout_s", {-1}}, {"pad_tiles", {-1}}};
}  // namespace
long long knob(const char* name) {
  for (Knob& k : g_knobs)
    if (!strcmp(k.name, name)) return k.value.load();
  return -1;
}

int require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(PGO_ERR_NO_DEVICE, "no HIP device visible (this backend has no CPU path)");
  if (device < 0 || device >= n) return fail(PGO_ERR_INVALID_ARG, "device index out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(PGO_ERR_HIP, "hipGetDeviceProperties");
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return fail(PGO_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
  return PGO_OK;
}

extern "C" {

void pgo_options_default(pgo_options* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->method = 1;
  o->max_iters = 50;
  o->fixed_pose = 0;
  o->jacobi_scaling = 1;
  o->phi = 0.5;
  o->huber_delta = 0.01;
  o->ftol = 1e-6;
  o->gtol = 1e-10;
  o->ptol = 1e-8;
  o->radius0 = 1e4;
  o->max_radius = 1e16;
  o->min_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->pcg_rtol = 1e-10;
  o->pcg_max_iters = 50000;
  o->pcg_check_every = 50;
  o->verbose = 0;
  o->use_graphs = 1;
  o->sc_prior_lambda = 1.0;
  o->pose_ordering = -1;
  o->pcg_chain_len = -1;
  o->halo_exchange = 0;  // all-gather (has run through RCCL, capturable); 1 = point-to-point exchange of the referenced rows:
                         // opt-in like halo_overlap until a multi-GPU run has checked it against the all-gather (bench.py does
                         // that check itself when it runs on several GPUs and then times the verified p2p path)
  o->halo_overlap = 0;   // opt-in: the two-stream schedule has never run against a real peer (no multi-GPU lease yet)
  o->linear_solver = 0;  // auto: the direct chain + low-rank solve on small chain-like graphs in the exact mode, else PCG
  o->pcg_coarse_poses = -1;  // auto: a rigid-body coarse level for exact-mode PCG solves of graphs of >= 512 poses
}

int pgo_create(pgo_t** h, int32_t n_poses, const double* poses, int32_t n_edges, const int32_t* ia, const int32_t* ib,
               const double* meas, const uint8_t* kind, const pgo_options* opt, pgo_comm* comm, int device) {
  return pgo_create_weighted(h, n_poses, poses, n_edges, ia, ib, meas, nullptr, kind, opt, comm, device);
}

int pgo_create_weighted(pgo_t** h, int32_t n_poses, const double* poses, int32_t n_edges, const int32_t* ia,
                        const int32_t* ib, const double* meas, const double* info6_or_null, const uint8_t* kind,
                        const pgo_options* opt, pgo_comm* comm, int device) {
  if (!h || !poses || n_poses <= 0 || n_edges < 0 || (n_edges && (!ia || !ib || !meas || !kind)))
    return fail(PGO_ERR_INVALID_ARG, "pgo_create: bad argument");
  pgo_options o;
  if (opt) o = *opt;
  else pgo_options_default(&o);
  if (o.method < 0 || o.method > 2)
    return fail(PGO_ERR_UNSUPPORTED, "only METHOD 0 (plain), 1 (DCS) and 2 (switchable constraints) are implemented (reference main.cpp:54-56)");
  if (o.fixed_pose >= n_poses) return fail(PGO_ERR_INVALID_ARG, "fixed_pose out of range");
  // every endpoint is checked HERE -- before the device is touched and before anything indexes by it
  // (resolve_chain_len, compute_pose_order and build_shard_structure all do)
  for (int32_t e = 0; e < n_edges; ++e) {
    if (ia[e] < 0 || ia[e] >= n_poses || ib[e] < 0 || ib[e] >= n_poses)
      return fail(PGO_ERR_INVALID_ARG, "edge " + std::to_string(e) + ": endpoint out of range");
    if (ia[e] == ib[e])
      return fail(PGO_ERR_INVALID_ARG, "edge " + std::to_string(e) + ": self loop (Ceres rejects duplicate parameter blocks)");
  }
  PGOC(require_device(device));
  std::unique_ptr<pgo_handle> H(new pgo_handle);
  H->opt = o;
  H->comm = comm;
  H->device = device;
  PGOC(H->create(n_poses, poses, n_edges, ia, ib, meas, info6_or_null, kind));
  *h = H.release();
  return PGO_OK;
}

int pgo_create_from_graph(pgo_t** h, const pgo_graph* g, const pgo_options* opt, pgo_comm* comm, int device) {
  if (!g) return fail(PGO_ERR_INVALID_ARG, "pgo_create_from_graph: null graph");
  const pgo::Graph& G = g->g;
  return pgo_create_weighted(h, G.n_poses(), G.pose.data(), G.n_edges(), G.ea.data(), G.eb.data(), G.meas.data(),
                             G.info.size() == (size_t)6 * G.n_edges() ? G.info.data() : nullptr, G.kind.data(), opt, comm, device);
}

void pgo_destroy(pgo_t* h) { delete h; }

int pgo_set_poses(pgo_t* h, const double* poses) {
  if (!h || !poses) return fail(PGO_ERR_INVALID_ARG, "pgo_set_poses: null");
  HIPC(hipSetDevice(h->device));
  std::vector<double> tmp;
  if (!h->perm.empty()) {
    h->to_internal(poses, &tmp, 3);
    poses = tmp.data();
  }
  HIPC(hipMemcpyAsync(h->poses, poses, (size_t)3 * h->S.n_poses * sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->lin_valid = false;
  h->lm_active = false;
  return h->sync();
}

int pgo_get_poses(pgo_t* h, double* out) {
  if (!h || !out) return fail(PGO_ERR_INVALID_ARG, "pgo_get_poses: null");
  HIPC(hipSetDevice(h->device));
  if (h->perm.empty()) {
    HIPC(hipMemcpyAsync(out, h->poses, (size_t)3 * h->S.n_poses * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return h->sync();
  }
  std::vector<double> tmp((size_t)3 * h->S.n_poses);
  HIPC(hipMemcpyAsync(tmp.data(), h->poses, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PGOC(h->sync());
  h->to_caller(tmp, out, 3);
  return PGO_OK;
}

int pgo_get_switches(pgo_t* h, double* switches, double* js_out) {
  if (!h || !switches) return fail(PGO_ERR_INVALID_ARG, "pgo_get_switches: null");
  if (h->comm && h->comm->world > 1) return fail(PGO_ERR_UNSUPPORTED, "pgo_get_switches: world == 1 only");
  HIPC(hipSetDevice(h->device));
  const int64_t EL = h->S.n_edges_local;
  if (!h->has_sw) {
    for (int64_t k = 0; k < EL; ++k) switches[h->S.orig_edge[k]] = 1.0;
    if (js_out) memset(js_out, 0, (size_t)3 * EL * sizeof(double));
    return PGO_OK;
  }
  std::vector<double> v((size_t)EL), j((size_t)3 * EL);
  HIPC(hipMemcpyAsync(v.data(), h->sw, v.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipMemcpyAsync(j.data(), h->sw_js, j.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PGOC(h->sync());
  for (int64_t k = 0; k < EL; ++k) {
    const int64_t e = h->S.orig_edge[k];
    const bool robust = h->S.flags[k] & 1;
    switches[e] = robust ? v[k] : 1.0;
    if (js_out)
      for (int c = 0; c < 3; ++c) js_out[3 * e + c] = robust ? j[3 * k + c] : 0.0;
  }
  return PGO_OK;
}

int pgo_eval(pgo_t* h, const double* poses_or_null, int apply_loss, double* cost, double* r_out, double* J_out) {
  if (!h) return fail(PGO_ERR_INVALID_ARG, "pgo_eval: null handle");
  HIPC(hipSetDevice(h->device));
  const bool want_jac = r_out || J_out;
  if (want_jac && h->comm && h->comm->world > 1) return fail(PGO_ERR_UNSUPPORTED, "pgo_eval: r/J outputs need world == 1");
  const double* x = h->poses;
  if (poses_or_null) {
    std::vector<double> tmp;
    const double* src = poses_or_null;
    if (!h->perm.empty()) {
      h->to_internal(poses_or_null, &tmp, 3);
      src = tmp.data();
    }
    HIPC(hipMemcpyAsync(h->cand, src, (size_t)3 * h->S.n_poses * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PGOC(h->sync());  // tmp dies with this scope
    x = h->cand;
  }
  if (want_jac) {
    h->lin_valid = false;  // the record buffer is about to be overwritten
    h->lm_active = false;
  }
  PGOC(h->eval_enqueue(x, h->sw, apply_loss, want_jac, 0));
  PGOC(h->fetch_scal(0, 2));
  if (cost) *cost = h->h_scal[0];
  if (want_jac) {
    const int64_t EL = h->S.n_edges_local;
    const int RN = h->rec_doubles, R0 = h->info_mode ? 12 : 10;
    std::vector<double> rec((size_t)EL * RN);
    HIPC(hipMemcpyAsync(rec.data(), h->jr, rec.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    PGOC(h->sync());
    for (int64_t k = 0; k < EL; ++k) {
      const int64_t e = h->S.orig_edge[k];
      const double* R = &rec[(size_t)k * RN];
      if (J_out) {  // expand the implied second block: d e/d P2 = [-A[:,0] | -A[:,1] | (0,0,g2)']
        double* Jo = J_out + 18 * e;
        for (int i = 0; i < 3; ++i) {
          Jo[6 * i + 0] = R[3 * i];
          Jo[6 * i + 1] = R[3 * i + 1];
          Jo[6 * i + 2] = R[3 * i + 2];
          Jo[6 * i + 3] = -R[3 * i];
          Jo[6 * i + 4] = -R[3 * i + 1];
          Jo[6 * i + 5] = h->info_mode ? R[9 + i] : ((i == 2) ? R[9] : 0.0);
        }
      }
      if (r_out) memcpy(r_out + 3 * e, R + R0, 3 * sizeof(double));
    }
  }
  if (h->h_scal[1] > 0.0) return fail(PGO_ERR_NUMERIC, "non-finite residual or Jacobian");
  return PGO_OK;
}

int pgo_edge_chi2(pgo_t* h, const double* poses_or_null, double* chi2_out) {
  if (!h || !chi2_out) return fail(PGO_ERR_INVALID_ARG, "pgo_edge_chi2: null");
  if (!h->e_info) return fail(PGO_ERR_INVALID_ARG, "pgo_edge_chi2: the handle was created without information matrices");
  HIPC(hipSetDevice(h->device));
  const int64_t E = h->n_edges_total, EL = h->S.n_edges_local;
  if (E == 0) return PGO_OK;
  if (!h->chi2_buf) {
    PGOC(h->dalloc(&h->chi2_buf, E));
    PGOC(h->dalloc(&h->e_orig, std::max<int64_t>(EL, 1)));
    PGOC(h->upload(h->e_orig, h->S.orig_edge));
  }
  const double* x = h->poses;
  std::vector<double> tmp;
  if (poses_or_null) {
    const double* src = poses_or_null;
    if (!h->perm.empty()) {
      h->to_internal(poses_or_null, &tmp, 3);
      src = tmp.data();
    }
    HIPC(hipMemcpyAsync(h->cand, src, (size_t)3 * h->S.n_poses * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PGOC(h->sync());
    x = h->cand;
  }
  HIPC(hipMemsetAsync(h->chi2_buf, 0, (size_t)E * sizeof(double), h->stream));
  if (EL > 0) {
    dev::EdgeArgs A = h->edge_args(x, nullptr, 0);
    const int grid = (int)std::min<int64_t>((EL + dev::WG - 1) / dev::WG, 8192);
    hipLaunchKernelGGL(dev::k_edge_chi2<>, dim3(grid), dim3(dev::WG), 0, h->stream, A, (const int32_t*)h->e_orig, h->chi2_buf);
    PGOC(h->check_launch("k_edge_chi2"));
  }
  if (h->multi_rank()) {  // every edge is counted on exactly one rank (flags bit1): the sum assembles the vector
    for (int64_t off = 0; off < E; off += (1 << 16)) {  // 512 KiB pieces (fits a slot of the shm test back-end)
      const int n = (int)std::min<int64_t>(E - off, 1 << 16);
      if (h->comm->allreduce(h->chi2_buf + off, n, false, h->stream) != 0) return fail(PGO_ERR_COMM, "pgo_edge_chi2: all-reduce failed");
    }
  }
  HIPC(hipMemcpyAsync(chi2_out, h->chi2_buf, (size_t)E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return h->sync();
}

int pgo_lm_begin(pgo_t* h) {
  if (!h) return fail(PGO_ERR_INVALID_ARG, "pgo_lm_begin: null handle");
  return h->lm_begin();
}

int pgo_lm_step(pgo_t* h, int32_t n_iters, int32_t* done, pgo_summary* s) {
  if (!h) return fail(PGO_ERR_INVALID_ARG, "pgo_lm_step: null handle");
  if (!h->lm_active || !h->lin_valid) return fail(PGO_ERR_INVALID_ARG, "pgo_lm_step: call pgo_lm_begin first");
  HIPC(hipSetDevice(h->device));
  bool stop = h->lm_done;
  for (int32_t k = 0; k < n_iters && !stop; ++k) PGOC(h->lm_iteration(&stop));
  h->lm_done = stop;
  if (done) *done = stop ? 1 : 0;
  h->fill_summary(s);
  return PGO_OK;
}

int pgo_solve(pgo_t* h, pgo_summary* s) {
  if (!h) return fail(PGO_ERR_INVALID_ARG, "pgo_solve: null handle");
  PGOC(h->lm_begin());
  bool stop = false;
  while (!stop) PGOC(h->lm_iteration(&stop));
  h->lm_done = true;
  h->fill_summary(s);
  return PGO_OK;
}

// Many independent solves at once (the reference's layer managers call ceres::Solve on a full-graph copy
// or a window per candidate layer / edge, src/simple_layer_manager.cpp:457-622, src/layer_manager.cpp:104-179): every
// handle owns its stream, device buffers and captured hipGraph, so `max_concurrency` host threads drive that many LM
// solves concurrently and the launch-/latency-bound small problems overlap on the device.
int pgo_solve_batch(pgo_t* const* handles, int32_t n, pgo_summary* summaries, int32_t max_concurrency) {
  if (n < 0 || (n > 0 && !handles)) return fail(PGO_ERR_INVALID_ARG, "pgo_solve_batch: bad argument");
  for (int32_t i = 0; i < n; ++i) {
    if (!handles[i]) return fail(PGO_ERR_INVALID_ARG, "pgo_solve_batch: null handle " + std::to_string(i));
    if (handles[i]->comm) return fail(PGO_ERR_UNSUPPORTED, "pgo_solve_batch: handles with a communicator solve collectively, one at a time");
    for (int32_t j = 0; j < i; ++j)
      if (handles[j] == handles[i]) return fail(PGO_ERR_INVALID_ARG, "pgo_solve_batch: handle " + std::to_string(i) + " listed twice");
  }
  if (n == 0) return PGO_OK;
  const int n_thr = std::max(1, std::min<int>(n, max_concurrency > 0 ? max_concurrency : 8));
  std::atomic<int32_t> next(0);
  std::mutex mu;
  int first_status = PGO_OK;
  std::string first_msg;
  auto worker = [&] {
    for (;;) {
      const int32_t i = next.fetch_add(1);
      if (i >= n) return;
      pgo_summary tmp;
      const int st = pgo_solve(handles[i], summaries ? &summaries[i] : &tmp);
      if (st != PGO_OK) {
        std::lock_guard<std::mutex> lk(mu);
        if (first_status == PGO_OK) {
          first_status = st;
          first_msg = "problem " + std::to_string(i) + ": " + pgo_last_error();  // the worker's thread-local text
        }
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_thr; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  if (first_status != PGO_OK) return fail(first_status, first_msg);
  return PGO_OK;
}

#ifdef PGO_PHASE_TIMING
// experiment builds only (scripts/exp_phase.sh): wall_clock64 stamps of the last k_cg_update1_cl launch
int pgo_debug_phase_times(pgo_t* h, unsigned long long* out16) {
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipMemcpyFromSymbol(out16, HIP_SYMBOL(dev::g_phase_t), 16 * sizeof(unsigned long long)));
  return PGO_OK;
}
#endif

int pgo_debug_set_knob(const char* name, long long value) {
  if (!name) return fail(PGO_ERR_INVALID_ARG, "pgo_debug_set_knob: null name");
  for (Knob& k : g_knobs)
    if (!strcmp(k.name, name)) {
      k.value.store(value < 0 ? -1 : value);
      return PGO_OK;
    }
  return fail(PGO_ERR_INVALID_ARG, std::string("pgo_debug_set_knob: unknown knob ") + name);
}

int pgo_get_info(const pgo_t* h, pgo_handle_info* out) {
  if (!h || !out) return fail(PGO_ERR_INVALID_ARG, "pgo_get_info: null");
  memset(out, 0, sizeof *out);
  out->n_poses = h->S.n_poses;
  out->n_edges = (int32_t)h->n_edges_total;
  out->world = h->comm ? h->comm->world : 1;
  out->rank = h->comm ? h->comm->rank : 0;
  out->row_lo = h->S.lo;
  out->row_hi = h->S.hi;
  out->n_edges_local = h->S.n_edges_local;
  out->n_tiles = h->S.n_tiles();
  out->n_incidences = h->S.n_inc_real;
  out->pcg_block_poses = h->grp_B;
  out->pcg_chain_len = h->chain_len;
  out->chain_kernel = h->chain_chunk;
  out->pose_ordering = h->perm.empty() ? 0 : 1;
  out->halo_exchange = h->use_halo ? 1 : 0;
  out->halo_overlap = h->overlap ? 1 : 0;
  out->halo_send_rows = (int64_t)h->S.halo_send_row.size();
  out->halo_recv_rows = (int64_t)h->S.halo_recv_row.size();
  out->device_bytes = h->device_bytes;
  out->host_enqueue_us_per_pcg_iter = h->n_enqueued > 0 ? 1e6 * h->t_enqueue / (double)h->n_enqueued : 0.0;
  out->pcg_graph_replay = (h->cg_graph_exec != nullptr && !h->graph_failed) ? 1 : 0;
  out->linear_solver = h->direct ? 2 : 1;
  out->direct_rank = h->direct ? h->dl_K : 0;
  out->direct_fallbacks = h->dl_fallbacks;
  out->direct_switched_at = h->dl_switched_at;
  out->pcg_single_reduction = h->use_sr ? 1 : 0;
  out->pcg_coarse_poses = h->use_coarse ? h->co_agg : 0;
  out->pcg_coarse_rank = h->use_coarse ? h->co_K : 0;
  return PGO_OK;
}

}  // extern "C"

extern "C" {

int32_t pgo_num_iter_records(const pgo_t* h) { return h ? (int32_t)h->recs.size() : 0; }
int pgo_get_iter_records(const pgo_t* h, pgo_iter_record* out, int32_t cap) {
  if (!h || !out) return fail(PGO_ERR_INVALID_ARG, "pgo_get_iter_records: null");
  int32_t n = std::min<int32_t>(cap, (int32_t)h->recs.size());
  memcpy(out, h->recs.data(), (size_t)n * sizeof(pgo_iter_record));
  return PGO_OK;
}

// ------------------------------------------------------------ debug / bench
int pgo_debug_normal_eq(pgo_t* h, double* g_out, double* hdiag_out) {
  if (!h) return fail(PGO_ERR_INVALID_ARG, "pgo_debug_normal_eq: null handle");
  if (h->comm && h->comm->world > 1) return fail(PGO_ERR_UNSUPPORTED, "world == 1 only");
  HIPC(hipSetDevice(h->device));
  h->lm_active = false;
  hipLaunchKernelGGL(dev::k_jacobi_scale<>, dim3(h->g_rows), dim3(dev::WG), 0, h->stream, h->hd, h->S.n_loc, h->S.lo,
                     h->fixed_internal, 0, h->scale, (const uint8_t*)h->fixed_mask);
  PGOC(h->check_launch("k_jacobi_scale"));
  int st = h->linearize(false);
  h->lin_valid = false;
  PGOC(st);
  const int64_t N = h->S.n_loc;
  std::vector<double> g_tmp;
  if (g_out) {
    g_tmp.resize((size_t)3 * N);
    HIPC(hipMemcpyAsync(g_tmp.data(), h->gs, g_tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  std::vector<double> planes;
  if (hdiag_out) {
    planes.resize((size_t)6 * N);
    HIPC(hipMemcpyAsync(planes.data(), h->hd, planes.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  PGOC(h->sync());
  auto internal = [&](int64_t i) { return h->perm.empty() ? i : (int64_t)h->perm[i]; };
  if (g_out)
    for (int64_t i = 0; i < N; ++i) memcpy(g_out + 3 * i, &g_tmp[(size_t)3 * internal(i)], 3 * sizeof(double));
  if (hdiag_out) {
    static const int map9[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    for (int64_t i = 0; i < N; ++i)
      for (int c = 0; c < 9; ++c) hdiag_out[9 * i + c] = planes[(size_t)map9[c] * N + internal(i)];
  }
  return PGO_OK;
}

int pgo_debug_spmv(pgo_t* h, const double* x, double* yout) {
  if (!h || !x || !yout) return fail(PGO_ERR_INVALID_ARG, "pgo_debug_spmv: null");
  if (h->comm && h->comm->world > 1) return fail(PGO_ERR_UNSUPPORTED, "world == 1 only");
  HIPC(hipSetDevice(h->device));
  const int64_t N = h->S.n_poses;
  std::vector<double> tmp;
  const double* src = x;
  if (!h->perm.empty()) {
    h->to_internal(x, &tmp, 3);
    src = tmp.data();
  }
  HIPC(hipMemcpyAsync(h->y, src, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(dev::k_scatter_owned<>, dim3(h->g_flat), dim3(dev::WG), 0, h->stream, h->S.n_loc, h->S.lo, h->y, h->p_full);
  PGOC(h->spmv_enqueue(h->p_full, h->ap, h->part[0], 0, nullptr));
  if (h->perm.empty()) {
    HIPC(hipMemcpyAsync(yout, h->ap, (size_t)3 * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    return h->sync();
  }
  std::vector<double> ytmp((size_t)3 * N);
  HIPC(hipMemcpyAsync(ytmp.data(), h->ap, ytmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PGOC(h->sync());
  h->to_caller(ytmp, yout, 3);
  return PGO_OK;
}

// z = M^-1 r with the preconditioner of the current LM iteration (whatever family the handle resolved to), through the
// PCG start-up kernel: for the symmetry / positivity property tests.  Needs at least one LM iteration; world == 1.
int pgo_debug_precond(pgo_t* h, const double* r_in, double* z_out) {
  if (!h || !r_in || !z_out) return fail(PGO_ERR_INVALID_ARG, "pgo_debug_precond: null");
  if (h->comm && h->comm->world > 1) return fail(PGO_ERR_UNSUPPORTED, "world == 1 only");
  if (!h->lin_valid || h->iter < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_debug_precond: run at least one LM iteration first");
  HIPC(hipSetDevice(h->device));
  const int64_t N = h->S.n_poses;
  std::vector<double> tmp;
  const double* src = r_in;
  if (!h->perm.empty()) {
    h->to_internal(r_in, &tmp, 3);
    src = tmp.data();
  }
  HIPC(hipMemcpyAsync(h->ap, src, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, h->stream));  // ap: scratch input
  if (h->direct) PGOC(h->prepare_preconditioner());   // (a handle on the direct solve does not factorise it per LM iteration)
  dev::CgVec V = h->cg_vec();
  if (h->chain_len) {
    h->launch_cg_init_chain(h->ap, h->part[0], h->part[1]);
  } else if (h->grp_B > 1) {
    dev::GroupPre GP;
    GP.ginv = h->ginv;
    GP.B = h->grp_B;
    GP.nb = h->grp_nb;
    GP.nb_pad = h->grp_pad;
    GP.n_groups = h->n_groups;
    hipLaunchKernelGGL(dev::k_cg_init_g<>, dim3(h->g_grp), dim3(dev::WG), 0, h->stream, V, GP, (const double*)h->ap, h->part[0], h->part[1]);
  } else {
    hipLaunchKernelGGL(dev::k_cg_init<>, dim3(h->g_vec), dim3(dev::WG), 0, h->stream, V, (const double*)h->ap, h->part[0], h->part[1]);
  }
  PGOC(h->check_launch("k_cg_init (debug)"));
  if (h->use_coarse) {   // the second level's share of z
    PGOC(h->coarse_solve(h->part[3], nullptr));
    hipLaunchKernelGGL(dev::k_coarse_prolong<>, dim3((unsigned)std::min<int64_t>((h->S.n_loc + 255) / 256, 512)), dim3(256), 0, h->stream,
                       (int)h->S.n_loc, h->co_agg, (const double*)h->co_pb, (const double*)h->co_ec, h->z, (double*)nullptr, (const int32_t*)h->co_ok);
    PGOC(h->check_launch("k_coarse_prolong"));
  }
  std::vector<double> ztmp((size_t)3 * N);
  HIPC(hipMemcpyAsync(ztmp.data(), h->z, ztmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PGOC(h->sync());
  if (h->perm.empty()) memcpy(z_out, ztmp.data(), ztmp.size() * sizeof(double));
  else h->to_caller(ztmp, z_out, 3);
  return PGO_OK;
}

static int time_launches(pgo_handle* h, int reps, const std::function<void()>& launch, double* ms_avg) {
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  launch();  // one untimed launch
  HIPC(hipEventRecord(e0, h->stream));
  for (int i = 0; i < reps; ++i) launch();
  HIPC(hipEventRecord(e1, h->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *ms_avg = (double)ms / reps;
  return h->check_launch("bench launch");
}

int pgo_bench_eval(pgo_t* h, int reps, int with_jacobian, pgo_kernel_stats* out) {
  if (!h || !out || reps < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_eval: bad argument");
  HIPC(hipSetDevice(h->device));
  double ms = 0;
  PGOC(time_launches(h, reps, [&] { h->launch_eval(h->poses, h->sw, 1, with_jacobian != 0); }, &ms));
  out->ms_avg = ms;
  out->units = h->S.n_edges_local;
  // 84 B read per edge + the record (here 112 B: DESIGN.md section 2); 84 + 8 without
  out->algorithmic_bytes = (double)h->S.n_edges_local * (with_jacobian ? 196.0 : 92.0);
  return PGO_OK;
}

int pgo_bench_assemble(pgo_t* h, int reps, pgo_kernel_stats* out) {
  if (!h || !out || reps < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_assemble: bad argument");
  if (!h->lin_valid) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_assemble: call pgo_lm_begin first");
  HIPC(hipSetDevice(h->device));
  double ms = 0;
  PGOC(time_launches(h, reps, [&] { (void)h->assemble_enqueue(); }, &ms));
  out->ms_avg = ms;
  out->units = h->S.n_edges_local;
  // every record read once (112 B/edge); per incidence 8 B indices + 72 B block written; per row 72 B out + 4 B
  // pointer + 24 B scale
  // chain preconditioner: + the 72-byte block (i, i-1) per row into the factorisation's input record
  out->algorithmic_bytes = 112.0 * h->S.n_edges_local + 80.0 * (double)h->S.n_inc_real + (100.0 + (h->chain_len ? 72.0 : 0.0)) * h->S.n_loc;
  return PGO_OK;
}

int pgo_bench_spmv(pgo_t* h, int reps, pgo_kernel_stats* out) {
  if (!h || !out || reps < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_spmv: bad argument");
  if (!h->lin_valid) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_spmv: call pgo_lm_begin first");
  HIPC(hipSetDevice(h->device));
  double ms = 0;
  const char* ab = PGO_EXP_ENV("PGO_SPMV_ABLATE");
  h->spmv_ablate = ab ? atoi(ab) : 0;
  int st_ab = PGO_OK;
#ifdef PGO_EXPERIMENTS
  // PGO_SPMV_PSTRIDE = 12 | 36: the product kernel on a copy of p spread over 96 / 288 bytes per pose (timing only; the
  // result is the same product) -- are the gathers served by the Infinity Cache or by HBM?
  const char* pstr = getenv("PGO_SPMV_PSTRIDE");
  const int stride = pstr ? atoi(pstr) : 0;
  if ((stride == 12 || stride == 36) && h->spmv_pipe) {
    double* big = nullptr;
    const int64_t nf = h->n_full;
    HIPC(hipMalloc((void**)&big, (size_t)nf * stride * sizeof(double)));
    HIPC(hipMemsetAsync(big, 0, (size_t)nf * stride * sizeof(double), h->stream));
    HIPC(hipMemcpy2DAsync(big, (size_t)stride * sizeof(double), h->p_full, 3 * sizeof(double), 3 * sizeof(double), (size_t)nf,
                          hipMemcpyDeviceToDevice, h->stream));
    dev::SpmvArgs A = h->spmv_args(big, h->ap, h->part[0], 1, nullptr);
    st_ab = time_launches(h, reps, [&] {
      if (stride == 12) hipLaunchKernelGGL(dev::k_spmv_p<12>, dim3(h->g_spmv), dim3(dev::WG), 0, h->stream, A);
      else hipLaunchKernelGGL(dev::k_spmv_p<36>, dim3(h->g_spmv), dim3(dev::WG), 0, h->stream, A);
    }, &ms);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(big);
  } else
#endif
  st_ab = time_launches(h, reps, [&] { (void)h->spmv_enqueue(h->p_full, h->ap, h->part[0], 1, nullptr); }, &ms);
  h->spmv_ablate = 0;
  PGOC(st_ab);
  out->ms_avg = ms;
  out->units = h->S.n_inc_real + h->S.n_loc;
  // 76 B per off-diagonal block (value + column) ; per row: 48 B diagonal planes + 24 B D'D + 4 B row
  // pointer + 24 B y + 24 B p; the product kernel k_spmv_p reads the diagonal with D'D folded in (k_prepare): 24 B less
  out->algorithmic_bytes = 76.0 * (double)h->S.n_inc_real + (h->spmv_pipe ? 100.0 : 124.0) * h->S.n_loc;
  return PGO_OK;
}

// the preconditioner apply as the PCG start-up kernel issues it (z = M^-1 b; writes y, r, z, p): timing only
int pgo_bench_precond(pgo_t* h, int reps, pgo_kernel_stats* out) {
  if (!h || !out || reps < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_precond: bad argument");
  if (!h->lin_valid || h->iter < 1) return fail(PGO_ERR_INVALID_ARG, "pgo_bench_precond: run at least one LM iteration first");
  HIPC(hipSetDevice(h->device));
  if (h->direct) PGOC(h->prepare_preconditioner());   // (a handle on the direct solve does not factorise it per LM iteration)
  dev::CgVec V = h->cg_vec();
  double ms = 0;
  const double nl = (double)h->S.n_loc;
  if (h->chain_len) {
    PGOC(time_launches(h, reps, [&] { h->launch_cg_init_chain(h->gs, h->part[0], h->part[1]); }, &ms));
    out->algorithmic_bytes = (120.0 + 24.0 + 4 * 24.0) * nl;   // W, S^-1 planes + b read; y, r, z, p written
  } else if (h->grp_B > 1) {
    dev::GroupPre GP;
    GP.ginv = h->ginv;
    GP.B = h->grp_B;
    GP.nb = h->grp_nb;
    GP.nb_pad = h->grp_pad;
    GP.n_groups = h->n_groups;
    PGOC(time_launches(h, reps, [&] {
      hipLaunchKernelGGL(dev::k_cg_init_g<>, dim3(h->g_grp), dim3(dev::WG), 0, h->stream, V, GP, (const double*)h->gs, h->part[0], h->part[1]);
    }, &ms));
    out->algorithmic_bytes = (8.0 * 3 * h->grp_nb + 24.0 + 4 * 24.0) * nl;
  } else {
    PGOC(time_launches(h, reps, [&] {
      hipLaunchKernelGGL(dev::k_cg_init<>, dim3(h->g_vec), dim3(dev::WG), 0, h->stream, V, (const double*)h->gs, h->part[0], h->part[1]);
    }, &ms));
    out->algorithmic_bytes = (48.0 + 24.0 + 4 * 24.0) * nl;
  }
  out->ms_avg = ms;
  out->units = h->S.n_loc;
  return PGO_OK;
}

}  // extern "C"

