#pragma once
// The solver handle of the pose-graph backend: device buffers, launch helpers and the declarations of the pieces that
// live in the translation units around it --
//   solver_create.hip   pgo_handle::create: shard structure -> device, solver / preconditioner choices (direct_setup, coarse_setup)
//   solver_lm.hip       Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy policy (lm_begin, lm_iteration, ...)
//   solver_pcg.hip      block-Jacobi PCG, its preconditioners' set-up, the coarse level
//   solver_direct.hip   the direct chain + low-rank solve
//   solver_batch.hip    pgo_batch: many independent problems in one handle
//   solver_abi.hip      the [gpu] part of the C-ABI, test hooks, debug / bench entry points
// Every kernel header declares its kernels `static`: a translation unit carries the kernels it launches.
//
// Replaces, for DCS-ceres/main.cpp METHOD 0/1 (paths relative to /root/reference/DCS-ceres):
//   main.cpp:66-68,95-153   problem assembly  -> pgo_create (shard structure + device upload)
//   main.cpp:154-163        ceres::Solve      -> pgo_solve / pgo_lm_begin + pgo_lm_step
// The minimiser follows Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy defaults
// (SURVEY.md R9); the linear solve is block-Jacobi PCG on the Jacobi-scaled normal equations.
//
// There is NO CPU fallback here: every [gpu] entry point fails with PGO_ERR_NO_DEVICE when no
// gfx950 device is visible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "comm.h"
#include "kernels.hip.h"
#include "solo.hip.h"
#include "direct.hip.h"
#include "coarse.hip.h"
#include "pgo_internal.h"

using pgo::fail;
namespace dev = pgo::dev;

#define HIPC(expr)                                                                       \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) return fail(PGO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
#define PGOC(expr)            \
  do {                        \
    int _s = (expr);          \
    if (_s != PGO_OK) return _s; \
  } while (0)

// Experiment switches (scripts/exp_*.sh build a library of their own with -DPGO_EXPERIMENTS and select it with PGO_LIB):
// the product library reads two documented environment variables only -- PGO_FORCE_COLLECTIVES, PGO_GRAPH_COLLECTIVES --
// and never lets the environment override a pgo_options field.
#ifdef PGO_EXPERIMENTS
#define PGO_EXP_ENV(name) getenv(name)
#else
#define PGO_EXP_ENV(name) ((const char*)nullptr)
#endif

// Test hooks (pgo_debug_set_knob, include/pgo.h): process-wide, read when a handle is created.  -1 = library default.
// (table and setter: solver_abi.hip)
long long knob(const char* name);
int require_device(int device);   // a gfx950 device of that index is visible (solver_abi.hip)

static inline double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct PartRef {   // one array of per-workgroup partials for reduce_to_scal
  const double* p;
  int n;
  int is_max;
};
namespace {
constexpr int N_SCAL = 16;
constexpr int N_PART = 6;
constexpr int DIRECT_MAX_POSES = 65536;   // largest graph the direct (chain + low-rank) solve takes
constexpr int DIRECT_MAX_RANK = 6144;    // 3 x (edges outside the chain) + 1: order of the dense capacitance matrix
// Cost model of the adaptive PCG / direct choice (solver_lm.hip): MEASURED ON MI355X -- a PCG iteration of the two-launch
// loop on graphs of a few thousand poses, and the rank up to which auto takes the direct solve outright (INTEL + 50: 918 ->
// 1.5 ms per LM iteration; FRH, 4515: 13 ms against 29 ms of PCG; M3500, 5862: 22 ms, the same as PCG).  Device-specific
// constants, not options: they only decide speed, never results.
constexpr double PCG_SECONDS_PER_ITER_SMALL = 14e-6;
constexpr int DIRECT_PROBE_EVERY = 10;
constexpr int DIRECT_AUTO_RANK = 2048;
constexpr int COARSE_MAX_RANK = 6144;        // order of the dense coarse matrix (k_chol_panel's range)
constexpr int COARSE_EXPLICIT_RANK = 1024;   // up to here the explicit inverse N'N is formed once per LM iteration
}  // namespace

struct pgo_handle {
  pgo_options opt;
  pgo::ShardStructure S;
  pgo_comm* comm = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  int64_t device_bytes = 0;

  int64_t n_full = 0;  // world * rows_per_rank  (>= N; tail rows are padding)
  // graph
  double *poses = nullptr, *cand = nullptr, *scale = nullptr;
  int32_t *e_ia = nullptr, *e_ib = nullptr;
  double *e_mx = nullptr, *e_my = nullptr, *e_mt = nullptr;
  uint8_t* e_flags = nullptr;
  double* jr = nullptr;
  // information matrices (6 planes over the local edges); info_mode = opt.info_weighting with them present
  double* e_info = nullptr;
  bool info_mode = false;
  int rec_doubles = dev::REC;
  int32_t* e_orig = nullptr;   // local edge -> caller's edge index (device copy, for pgo_edge_chi2)
  double* chi2_buf = nullptr;  // [n_edges_total], allocated at the first pgo_edge_chi2
  int64_t n_edges_total = 0;
  int32_t *inc_ptr = nullptr, *inc_edge = nullptr, *inc_col = nullptr, *tile_row = nullptr;
  uint8_t* inc_rowoff = nullptr;
  int4* tile_desc = nullptr;
  bool spmv_one_tile = false;   // the plain-tile product kernel as k_spmv_1: one tile per workgroup, as many workgroups as tiles
  bool spmv_pipe = false;   // software-pipelined K3 (k_spmv_p): when no tile is a chunked heavy row or has > 85 rows
  int64_t inc_stride = 0;
  // normal equations
  double *hoff = nullptr, *hd = nullptr, *gs = nullptr, *d2 = nullptr, *minv = nullptr, *hdd = nullptr;
  // CG
  double *y = nullptr, *r = nullptr, *z = nullptr, *ap = nullptr, *p_full = nullptr;
  // second preconditioner level (coarse.hip.h): additive coarse correction on the rigid-body modes of pose aggregates
  bool use_coarse = false;
  int co_agg = 0, co_nagg = 0, co_K = 0, co_Kp = 0, co_ncb = 0;
  double *co_pb = nullptr, *co_cap = nullptr, *co_nm = nullptr, *co_dwork = nullptr, *co_rc = nullptr, *co_cy = nullptr, *co_ec = nullptr;
  int32_t *co_cb_i = nullptr, *co_cb_j = nullptr, *co_cb_ptr = nullptr, *co_cb_q = nullptr, *co_cb_row = nullptr;
  int coarse_setup();      // create: aggregates, coarse block lists, buffers
  int coarse_factor();     // per LM iteration: basis, Galerkin matrix, Cholesky + inverse factor
  double* co_ainv = nullptr;   // explicit inverse N'N (coarse orders <= COARSE_EXPLICIT_RANK: one product per apply)
  int32_t* co_ok = nullptr;    // device flag: the factorisation of this LM iteration is usable
  int co_ndot = 0;             // partials of r_c . e_c appended to the r.z partials
  int coarse_solve(double* dot_part, const int32_t* done);   // e_c = (P'(H + D'D)P)^-1 P' r  (+ partials of r_c . e_c)
  // single-reduction PCG loop (k_cg_sr_*: one all-reduce per iteration; several ranks, inexact mode, chain preconditioner)
  bool use_sr = false;
  bool verify_residual = false;   // test hook: pcg() reports the true residual of its solution
  double* sr_s = nullptr;   // s = A p, carried by recurrence
  dev::CgState* st = nullptr;
  dev::CgState* h_st = nullptr;  // pinned
  // reductions
  double* part[N_PART] = {nullptr};
  double* fold_buf = nullptr;   // 6 x 16: reduce_to_scal's intermediate for partial arrays too long for one workgroup
  int part_cap = 0;
  double* scal = nullptr;
  double* h_scal = nullptr;  // pinned
  int* bad = nullptr;
  // block-Jacobi over groups of B poses (B > 1): explicit dense inverses
  int grp_B = 1, grp_nb = 3, grp_pad = 32, n_groups = 0, g_grp = 1;
  size_t grp_lds = 0;
  int grp_prep_grid = 1;
  double* ginv = nullptr;
  // chain (block-tridiagonal) preconditioner over 64-pose segments (opt.pcg_chain_len): C planes, W planes, S^-1 planes
  int chain_len = 0, g_chain = 1, chain_pad = 0;
  int chain_chunk = 0, chain_steps = 0;  // lean apply: poses per lane (2 / 4) and recurrence steps; 0 = scan kernel (layout chunk 4)
  int chain_scan = 0;                    // > 0: the lean apply runs its recurrence as this many scan levels (few long segments)
  int chain_nw = 4;                      // wavefronts per workgroup of the lean kernels: 1 on small graphs (a tile per CU)
  double *chain_c = nullptr, *chain_w = nullptr, *chain_s = nullptr;
  int32_t* chain_dup_rows = nullptr;   // rows whose block (i, i-1) sums several edges (k_chain_dupfix)
  int n_chain_dup = 0;
  // halo exchange of the search direction (world > 1, opt.halo_exchange)
  bool use_halo = false;
  int32_t *halo_send_rows = nullptr, *halo_recv_rows = nullptr;
  double *halo_send_buf = nullptr, *halo_recv_buf = nullptr;
  std::vector<int64_t> halo_send_off3, halo_recv_off3;  // offsets in doubles (3 per row)
  // overlap of the halo exchange with the SpMV (opt.halo_overlap): the blocks with owned columns (MODE 4 of k_spmv) are
  // multiplied while the exchange runs on a second stream; the few blocks with remote columns follow (k_spmv_remote)
  bool overlap = false;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
  int32_t *rr_rows = nullptr, *rr_ptr = nullptr, *rr_slots = nullptr;
  int n_rr = 0, g_spmv_loc = 0, g_rr = 0;
  // METHOD 2: switch variables (one per local edge; only robust edges use theirs), eliminated per edge
  bool has_sw = false;
  double *sw = nullptr, *sw_cand = nullptr, *sw_js = nullptr, *sw_sigma = nullptr, *sw_c = nullptr, *sw_gamma = nullptr,
         *sw_gs = nullptr, *sw_den = nullptr, *sw_hss = nullptr, *diag_full = nullptr, *gs_full = nullptr;
  double sw_norm2 = 0.0, xnorm2_pose = 0.0;
  bool sw_fresh = false;  // elimination coefficients / reduced system match the current point AND radius
  // internal pose numbering (opt.pose_ordering): perm[i] = internal position of the caller's pose i; empty = identity
  std::vector<int32_t> perm;
  int fixed_internal = -1;  // opt.fixed_pose in the internal numbering
  // host <-> device pose-vector helpers honouring the permutation (n_cols doubles per pose)
  void to_internal(const double* in, std::vector<double>* out, int n_cols) const {
    const int64_t N = S.n_poses;
    out->resize((size_t)N * n_cols);
    for (int64_t i = 0; i < N; ++i) memcpy(&(*out)[(size_t)perm[i] * n_cols], in + i * n_cols, (size_t)n_cols * sizeof(double));
  }
  void to_caller(const std::vector<double>& in, double* out, int n_cols) const {
    const int64_t N = S.n_poses;
    for (int64_t i = 0; i < N; ++i) memcpy(out + i * n_cols, &in[(size_t)perm[i] * n_cols], (size_t)n_cols * sizeof(double));
  }
  // captured slice of PCG iterations (world == 1)
  hipGraphExec_t cg_graph_exec = nullptr;
  int cg_graph_len = 0;
  bool graph_failed = false;
  int graph_collectives = 1;   // PGO_GRAPH_COLLECTIVES: 0 = never capture collectives, 1 = all-reduce / all-gather, 2 = also the p2p halo exchange
  int last_pcg_iters = 0;  // iteration count of the previous PCG solve of this handle (slice scheduling)
  double t_enqueue = 0.0;  // host seconds spent enqueueing PCG iterations (launch calls only, no waiting), and how many
  int64_t n_enqueued = 0;
  // small graphs on one rank: the direction update rides in the next SpMV (k_spmv MODE 5) -- two launches per PCG
  // iteration instead of three; p_full / p_full2 alternate by iteration parity
  bool fused_p = false;
  double* p_full2 = nullptr;
  // batched handle (pgo_batch_*): the block-diagonal union of independent problems, each starting at a multiple of 256 rows
  bool batch_mode = false;
  std::vector<uint8_t> fixed_mask_h;    // set before create(): constant rows (one anchor per problem + the padding rows)
  std::vector<int32_t> tile_breaks_h;   // set before create(): rows at which a row tile must start (problem starts)
  uint8_t* fixed_mask = nullptr;
  int32_t* prob_of_256 = nullptr;       // problem of each 256-row block
  double* prob_radius = nullptr;        // trust-region radius per problem
  double* edge_cost = nullptr;          // cost per local edge (k_edge_eval -> k_prob_reduce)
  // small graphs: the whole PCG solve of an LM iteration as ONE launch, one workgroup (solo.hip.h)
  bool solo = false;
  dev::SoloProb* solo_prob = nullptr;
  dev::SoloOut* solo_out = nullptr;
  dev::SoloOut* h_solo = nullptr;  // pinned
  int solo_steps = 0, solo_scan = 0;
  // grids
  int g_edge = 1, g_rows = 1, g_vec = 1, g_flat = 1, g_spmv = 1, g_asm = 1;
  // direct solve for small chain-like graphs (direct.hip.h): T (odometry chain) + V'V (the other edges) by Woodbury
  bool direct = false;
  int dl_m = 0, dl_K = 0, dl_Kp = 0, dl_ld = 0, dl_refine = 1;
  int32_t *dl_chain_edge = nullptr, *dl_lr_edge = nullptr, *dl_va = nullptr, *dl_vb = nullptr;
  double *dl_trec = nullptr, *dl_fac = nullptr, *dl_pre = nullptr, *dl_vrec = nullptr, *dl_Z = nullptr, *dl_cap = nullptr, *dl_dwork = nullptr,
         *dl_nm = nullptr, *dl_cy = nullptr, *dl_cvec = nullptr, *dl_x1 = nullptr, *dl_E = nullptr, *dl_E2 = nullptr;
  int dl_nseg = 1, dl_seglen = 1;
  int dl_nseg2 = 1, dl_seglen2 = 1;   // the finer segmentation of k_dlr_solve1 (up to 256 segments of <= 16 poses)
  double* dl_pre2 = nullptr;
  int dl_nsep = 0, dl_sep[dev::DLR_MAX_SEP] = {0}, dl_nU = 0;
  double *dl_ksep = nullptr, *dl_R = nullptr, *dl_Wm = nullptr;
  double dl_rel = 0.0;  // |g - (H + D'D) y| / |g| of the latest direct solve
  bool dl_retry = false;   // the current LM iteration is being redone by PCG after a failed direct solve
  int dl_fallbacks = 0;    // how often that happened
  int dl_fail_at = 0;      // test hook (pgo_debug_set_knob "direct_fail_at"): poison the direct solve of this LM iteration
  bool dl_possible = false;     // auto, rank above DIRECT_AUTO_RANK: the direct solve can take over from PCG (lm_iteration)
  double dl_est_seconds = 0.0;  // what a direct solve of this rank costs (model fitted to INTEL / FRH / M3500)
  int dl_switched_at = 0;       // LM iteration after which it first did
  int dl_last_probe = 0, dl_dear_run = 0;
  bool dl_ready = false;        // the direct solve's buffers exist
  void clear_direct_buffers() {   // after a failed direct_setup(): every pointer it may have set (the memory is freed by the caller)
    dl_chain_edge = dl_lr_edge = dl_va = dl_vb = nullptr;
    dl_trec = dl_fac = dl_pre = dl_vrec = dl_Z = dl_cap = dl_dwork = dl_nm = dl_cy = dl_cvec = dl_x1 = dl_E = dl_E2 = nullptr;
    dl_pre2 = dl_ksep = dl_R = dl_Wm = nullptr;
    dl_ready = false;
  }
  hipGraphExec_t dl_graph_exec = nullptr;   // the captured direct solve
  bool dl_graph_failed = false;
  bool dl_use_graph = false;   // PGO_DIRECT_GRAPH=1

  // LM state (TrustRegionMinimizer)
  bool lm_active = false, lin_valid = false, lm_done = false;
  int iter = 0, prev_success = 1, invalid_run = 0, successful = 0, total_pcg = 0, termination = 0;
  double cost = 0, initial_cost = 0, radius = 0, decrease_factor = 2, x_norm = 0, gmax = 0;
  double t_eval = 0, t_asm = 0, t_lin = 0, t_cand = 0, t_total = 0;
  std::vector<pgo_iter_record> recs;

  ~pgo_handle() {
    if (device >= 0) (void)hipSetDevice(device);
    if (ev_pack) (void)hipEventDestroy(ev_pack);
    if (ev_halo) (void)hipEventDestroy(ev_halo);
    if (comm_stream) (void)hipStreamDestroy(comm_stream);
    for (void* p : allocs) (void)hipFree(p);
    if (cg_graph_exec) (void)hipGraphExecDestroy(cg_graph_exec);
    if (dl_graph_exec) (void)hipGraphExecDestroy(dl_graph_exec);
    if (h_st) (void)hipHostFree(h_st);
    if (h_scal) (void)hipHostFree(h_scal);
    if (h_solo) (void)hipHostFree(h_solo);
    if (stream) (void)hipStreamDestroy(stream);
  }

  // collectives are skipped for a single rank unless PGO_FORCE_COLLECTIVES=1 (lets a 1-GPU box
  // exercise the RCCL calls themselves: at world == 1 they are identities)
  bool force_collectives = false;
  int spmv_ablate = 0;  // timing-only ablations of k_spmv, set by pgo_bench_spmv from PGO_SPMV_ABLATE
  int spmv_nt = 1;      // non-temporal H-stream loads in k_spmv (PGO_SPMV_NT=0 turns them off): 179 -> 166 us at 1M poses
  bool multi_rank() const { return comm && (comm->world > 1 || force_collectives); }

  template <class T>
  int dalloc(T** out, int64_t n) {
    void* p = nullptr;
    size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
    // (one hipMalloc per buffer on purpose: all buffers of a handle carved out of ONE allocation were measured 5-12 % slower in
    // K1 / K2 / K3 -- DESIGN.md section 8, round 3 (g))
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(PGO_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    allocs.push_back(p);
    device_bytes += (int64_t)bytes;
    e = hipMemsetAsync(p, 0, bytes, stream);
    if (e != hipSuccess) return fail(PGO_ERR_HIP, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
    *out = (T*)p;
    return PGO_OK;
  }
  template <class T>
  int upload(T* dst, const std::vector<T>& src) {
    if (src.empty()) return PGO_OK;
    HIPC(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return PGO_OK;
  }
  int sync() {
    HIPC(hipStreamSynchronize(stream));
    return PGO_OK;
  }
  int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PGO_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return PGO_OK;
  }

  // ---- reductions to scalars: scal[first..first+k) = reduce(parts) [+ all-reduce], no host sync
  int reduce_to_scal(std::initializer_list<PartRef> parts, int first, bool allreduce_max = false);
  int fetch_scal(int first, int count) {
    HIPC(hipMemcpyAsync(h_scal + first, scal + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, stream));
    return sync();
  }
  int allgather(double* full, int stride = 3) {
    if (multi_rank()) PGOC(comm->allgather_inplace(full, (int64_t)stride * S.rows_per_rank, stream));
    return PGO_OK;
  }
  // make the owned rows of the gather vector visible where the peers need them: either everything
  // (all-gather) or only the rows their off-diagonal blocks reference (halo exchange)
  int share_gather_vector(double* full);

  // ---- K1
  dev::EdgeArgs edge_args(const double* x, const double* sw_vals, int apply_loss) const {
    dev::EdgeArgs A;
    A.poses = x;
    A.ia = e_ia;
    A.ib = e_ib;
    A.mx = e_mx;
    A.my = e_my;
    A.mt = e_mt;
    A.flags = e_flags;
    A.n_edges = S.n_edges_local;
    A.apply_loss = apply_loss;
    A.phi = opt.phi;
    A.huber_delta = opt.huber_delta;
    A.sw = has_sw ? sw_vals : nullptr;
    A.sw_js = sw_js;
    A.sc_lambda = opt.sc_prior_lambda;
    A.info = e_info;
    A.cost_out = edge_cost;
    return A;
  }
  dev::SwitchArrays switch_arrays() const {
    dev::SwitchArrays W;
    W.flags = e_flags;
    W.n_edges = S.n_edges_local;
    W.lambda = opt.sc_prior_lambda;
    W.sw = sw;
    W.cand = sw_cand;
    W.js = sw_js;
    W.sigma = sw_sigma;
    W.c = sw_c;
    W.gamma = sw_gamma;
    W.gs = sw_gs;
    W.den = sw_den;
    W.hss = sw_hss;
    return W;
  }
  void launch_eval(const double* x, const double* sw_vals, int apply_loss, bool with_jac);
  // evaluates at x; on return h_scal[slot] = cost, h_scal[slot+1] = #bad flags (needs fetch by caller)
  int eval_enqueue(const double* x, const double* sw_vals, int apply_loss, bool with_jac, int slot);

  // ---- K2
  dev::AsmArgs asm_args() const {
    dev::AsmArgs A;
    A.jr = jr;
    A.inc_ptr = inc_ptr;
    A.inc_edge = inc_edge;
    A.inc_col = inc_col;
    A.tile_row = tile_row;
    A.inc_rowoff = inc_rowoff;
    A.tile_desc = tile_desc;
    A.scale = scale;
    A.n_tiles = S.n_tiles();
    A.n_loc = S.n_loc;
    A.lo = S.lo;
    A.inc_stride = inc_stride;
    A.hoff = hoff;
    A.hd = hd;
    A.gs = gs;
    A.sw_js = sw_js;
    A.sw_c = sw_c;
    A.sw_gamma = sw_gamma;
    A.diag_full = diag_full;
    A.gs_full = gs_full;
    A.chain_rec = chain_len ? chain_c : nullptr;
    A.chain_seg = chain_len ? chain_len : 1;
    return A;
  }
  int assemble_enqueue();

  // ---- K3
  dev::SpmvArgs spmv_args(const double* p, double* yout, double* dot_part, int with_d2, const int32_t* done) const {
    dev::SpmvArgs A;
    A.inc_ptr = inc_ptr;
    A.inc_col = inc_col;
    A.tile_row = tile_row;
    A.tile_desc = tile_desc;
    A.n_tiles = S.n_tiles();
    A.n_loc = S.n_loc;
    A.lo = S.lo;
    A.with_d2 = with_d2;
    A.nt = spmv_nt;
    A.inc_stride = inc_stride;
    A.hoff = hoff;
    A.hd = hd;
    A.d2 = d2;
    A.hdd = hdd;
    A.p = p;
    A.y = yout;
    A.dot_part = dot_part;
    A.done = done;
    A.z = nullptr;
    A.p_new = nullptr;
    A.part_rz = A.part_rr = nullptr;
    A.n_rz = A.n_rr = A.parity = 0;
    A.st = nullptr;
    return A;
  }
  int spmv_enqueue(const double* p, double* yout, double* dot_part, int with_d2, const int32_t* done);

  // PCG step "make p visible to the peers, then A p" with the halo exchange hidden behind the interior tiles.
  // *n_part = number of dot partials written to dot_part.
  int spmv_with_halo(double* full, double* yout, double* dot_part, const int32_t* done, int* n_part);

  dev::CgVec cg_vec() const {
    dev::CgVec V;
    V.n_loc = S.n_loc;
    V.lo = S.lo;
    V.minv = minv;
    V.y = y;
    V.r = r;
    V.z = z;
    V.ap = ap;
    V.p = p_full;
    V.st = st;
    V.fused = 0;
    V._pad = 0;
    return V;
  }

  dev::ChainPre chain_pre() const {
    dev::ChainPre CP;
    CP.cw = chain_w;
    CP.cs = chain_s;
    CP.n_loc = S.n_loc;
    CP.n_pad = chain_pad;
    return CP;
  }
  // PCG start-up / first update kernel with the chain preconditioner (scan or lean form)
  void launch_cg_init_chain(const double* b, double* part_rz, double* part_bb);
  void launch_cg_sr_chain(const dev::CgVec& V, double* part_gamma, double* part_rr);
  void launch_cg_update1_chain(const dev::CgVec& V, int par, const double* pap, int n_pap, double* part_rz, double* part_rr);

  int create(int32_t N, const double* poses_h, int32_t E, const int32_t* ia, const int32_t* ib, const double* meas,
             const double* info6, const uint8_t* kind);
  int linearize(bool reuse_records, bool assemble = true);
  int refresh_switch_system();
  int lm_begin();
  int lm_iteration(bool* stop);
  int lm_iteration_tail(bool* stop, pgo_iter_record& R, double it0, double t0, int k_it, double rel);
  int prepare_system();
  int pcg(int* iters, double* rel);
  int direct_setup(int32_t N, bool switch_now = false);
  int direct_solve();
  int direct_enqueue();
  int factor_chain();
  int prepare_preconditioner();
  void fill_summary(pgo_summary* s) const;
};

