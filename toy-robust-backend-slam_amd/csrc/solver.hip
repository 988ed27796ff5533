// LM + block-Jacobi PCG driver of the pose-graph backend and the [gpu] part of the C-ABI.
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
namespace {
struct Knob {
  const char* name;
  std::atomic<long long> value;
};
Knob g_knobs[] = {{"spmv_pipe", {-1}}, {"fused_p", {-1}}, {"direct_fail_at", {-1}}, {"direct_setup_fail", {-1}}, {"single_reduction", {-1}}};
long long knob(const char* name) {
  for (Knob& k : g_knobs)
    if (!strcmp(k.name, name)) return k.value.load();
  return -1;
}
}  // namespace

static double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace {
struct PartRef {
  const double* p;
  int n;
  int is_max;
};
constexpr int N_SCAL = 16;
constexpr int N_PART = 6;
constexpr int DIRECT_MAX_POSES = 65536;   // largest graph the direct (chain + low-rank) solve takes
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
  double* sr_s = nullptr;   // s = A p, carried by recurrence
  dev::CgState* st = nullptr;
  dev::CgState* h_st = nullptr;  // pinned
  // reductions
  double* part[N_PART] = {nullptr};
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
  int reduce_to_scal(std::initializer_list<PartRef> parts, int first, bool allreduce_max = false) {
    dev::FinArgs F;
    memset(&F, 0, sizeof F);
    int k = 0;
    for (const PartRef& pr : parts) {
      F.part[k] = pr.p;
      F.n[k] = pr.n;
      F.is_max[k] = pr.is_max;
      ++k;
    }
    F.count = k;
    F.out = scal + first;
    hipLaunchKernelGGL(dev::k_finalize, dim3(1), dim3(dev::WG), 0, stream, F);
    PGOC(check_launch("k_finalize"));
    if (multi_rank()) PGOC(comm->allreduce(scal + first, k, allreduce_max, stream));
    return PGO_OK;
  }
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
  int share_gather_vector(double* full) {
    if (!multi_rank()) return PGO_OK;
    if (!use_halo) return allgather(full, dev::PS);
    const int64_t ns = (int64_t)S.halo_send_row.size(), nr = (int64_t)S.halo_recv_row.size();
    if (ns > 0) {
      hipLaunchKernelGGL(dev::k_pack_rows, dim3((unsigned)std::min<int64_t>((3 * ns + 255) / 256, 2048)), dim3(256), 0, stream, ns,
                         (const int32_t*)halo_send_rows, (const double*)full, halo_send_buf);
      PGOC(check_launch("k_pack_rows"));
    }
    PGOC(comm->exchange(halo_send_buf, halo_send_off3.data(), halo_recv_buf, halo_recv_off3.data(), stream));
    if (nr > 0) {
      hipLaunchKernelGGL(dev::k_unpack_rows, dim3((unsigned)std::min<int64_t>((3 * nr + 255) / 256, 2048)), dim3(256), 0, stream, nr,
                         (const int32_t*)halo_recv_rows, (const double*)halo_recv_buf, full);
      PGOC(check_launch("k_unpack_rows"));
    }
    return PGO_OK;
  }

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
  void launch_eval(const double* x, const double* sw_vals, int apply_loss, bool with_jac) {
    dev::EdgeArgs A = edge_args(x, sw_vals, apply_loss);
    if (info_mode) {
      if (with_jac) hipLaunchKernelGGL((dev::k_edge_eval<true, true>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
      else hipLaunchKernelGGL((dev::k_edge_eval<false, true>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
    } else {
      if (with_jac) hipLaunchKernelGGL((dev::k_edge_eval<true, false>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
      else hipLaunchKernelGGL((dev::k_edge_eval<false, false>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
    }
  }
  // evaluates at x; on return h_scal[slot] = cost, h_scal[slot+1] = #bad flags (needs fetch by caller)
  int eval_enqueue(const double* x, const double* sw_vals, int apply_loss, bool with_jac, int slot) {
    HIPC(hipMemsetAsync(bad, 0, sizeof(int), stream));
    launch_eval(x, sw_vals, apply_loss, with_jac);
    PGOC(check_launch("k_edge_eval"));
    // the flag rides along as a "partial array" of length 1 after conversion to double
    hipLaunchKernelGGL(dev::k_flag_to_double, dim3(1), dim3(1), 0, stream, bad, part[4]);
    return reduce_to_scal({{part[5], g_edge, 0}, {part[4], 1, 0}}, slot);
  }

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
  int assemble_enqueue() {
    if (S.n_tiles() == 0) return PGO_OK;
    if (has_sw) hipLaunchKernelGGL((dev::k_assemble<true, false>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
    else if (info_mode) hipLaunchKernelGGL((dev::k_assemble<false, true>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
    else hipLaunchKernelGGL((dev::k_assemble<false, false>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
    PGOC(check_launch("k_assemble"));
    if (chain_len && n_chain_dup > 0) {
      hipLaunchKernelGGL(dev::k_chain_dupfix, dim3((n_chain_dup + 63) / 64), dim3(64), 0, stream, (const int32_t*)chain_dup_rows, n_chain_dup,
                         (const int32_t*)inc_ptr, (const int32_t*)inc_col, (const double*)hoff, S.lo, chain_c);
      PGOC(check_launch("k_chain_dupfix"));
    }
    return PGO_OK;
  }

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
  int spmv_enqueue(const double* p, double* yout, double* dot_part, int with_d2, const int32_t* done) {
    dev::SpmvArgs A = spmv_args(p, yout, dot_part, with_d2, done);
    switch (spmv_ablate) {
#ifdef PGO_EXPERIMENTS
      case 1: hipLaunchKernelGGL(dev::k_spmv_t<1>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
      case 2: hipLaunchKernelGGL(dev::k_spmv_t<2>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
      case 3: hipLaunchKernelGGL(dev::k_spmv_t<3>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
#endif
      default:
        if (spmv_pipe) hipLaunchKernelGGL(dev::k_spmv_p<dev::PS>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
        else hipLaunchKernelGGL(dev::k_spmv_t<0>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
    }
    return check_launch("k_spmv");
  }

  // PCG step "make p visible to the peers, then A p" with the halo exchange hidden behind the interior tiles.
  // *n_part = number of dot partials written to dot_part.
  int spmv_with_halo(double* full, double* yout, double* dot_part, const int32_t* done, int* n_part) {
    if (!overlap) {
      PGOC(share_gather_vector(full));
      *n_part = g_spmv;
      return spmv_enqueue(full, yout, dot_part, 1, done);
    }
    const int64_t ns = (int64_t)S.halo_send_row.size(), nr = (int64_t)S.halo_recv_row.size();
    if (ns > 0) {
      hipLaunchKernelGGL(dev::k_pack_rows, dim3((unsigned)std::min<int64_t>((3 * ns + 255) / 256, 2048)), dim3(256), 0, stream, ns,
                         (const int32_t*)halo_send_rows, (const double*)full, halo_send_buf);
      PGOC(check_launch("k_pack_rows"));
    }
    HIPC(hipEventRecord(ev_pack, stream));
    int used = 0;
    if (S.n_tiles() > 0) {  // enqueued before the exchange so that it also overlaps a host-blocking back-end
      dev::SpmvArgs A = spmv_args(full, yout, dot_part, 1, done);
      hipLaunchKernelGGL(dev::k_spmv_t<4>, dim3(g_spmv_loc), dim3(dev::WG), 0, stream, A);
      PGOC(check_launch("k_spmv (owned columns)"));
      used += g_spmv_loc;
    }
    HIPC(hipStreamWaitEvent(comm_stream, ev_pack, 0));
    PGOC(comm->exchange(halo_send_buf, halo_send_off3.data(), halo_recv_buf, halo_recv_off3.data(), comm_stream));
    if (nr > 0) {
      hipLaunchKernelGGL(dev::k_unpack_rows, dim3((unsigned)std::min<int64_t>((3 * nr + 255) / 256, 2048)), dim3(256), 0, comm_stream, nr,
                         (const int32_t*)halo_recv_rows, (const double*)halo_recv_buf, full);
      PGOC(check_launch("k_unpack_rows"));
    }
    HIPC(hipEventRecord(ev_halo, comm_stream));
    HIPC(hipStreamWaitEvent(stream, ev_halo, 0));
    if (n_rr > 0) {
      dev::RemoteArgs R;
      R.rows = rr_rows;
      R.ptr = rr_ptr;
      R.slots = rr_slots;
      R.inc_col = inc_col;
      R.hoff = hoff;
      R.p = full;
      R.y = yout;
      R.dot_part = dot_part + used;
      R.n_rows = n_rr;
      R.lo = S.lo;
      R.done = done;
      hipLaunchKernelGGL(dev::k_spmv_remote, dim3(g_rr), dim3(dev::WG), 0, stream, R);
      PGOC(check_launch("k_spmv_remote"));
      used += g_rr;
    }
    *n_part = used;
    return PGO_OK;
  }

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
  void launch_cg_init_chain(const double* b, double* part_rz, double* part_bb) {
    const dev::CgVec V = cg_vec();
    const dev::ChainPre CP = chain_pre();
    if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
    else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_init_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
    else if (chain_chunk == 4 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
    else if (chain_chunk == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
    else hipLaunchKernelGGL(dev::k_cg_init_c, dim3(g_chain), dim3(dev::WG), 0, stream, V, CP, b, part_rz, part_bb);
  }
  void launch_cg_sr_chain(const dev::CgVec& V, double* part_gamma, double* part_rr) {
    const dev::ChainPre CP = chain_pre();
    if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_sr_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
    else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_sr_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
    else if (chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_sr_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
    else hipLaunchKernelGGL((dev::k_cg_sr_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
  }
  void launch_cg_update1_chain(const dev::CgVec& V, int par, const double* pap, int n_pap, double* part_rz, double* part_rr) {
    const dev::ChainPre CP = chain_pre();
    if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
    else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_update1_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
    else if (chain_chunk == 4 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
    else if (chain_chunk == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
    else hipLaunchKernelGGL(dev::k_cg_update1_c, dim3(g_chain), dim3(dev::WG), 0, stream, V, CP, par, pap, n_pap, part_rz, part_rr);
  }

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

// --------------------------------------------------------------------- create
int pgo_handle::create(int32_t N, const double* poses_h, int32_t E, const int32_t* ia, const int32_t* ib,
                       const double* meas, const double* info6, const uint8_t* kind) {
  const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
  n_edges_total = E;
  info_mode = opt.info_weighting != 0;
  if (info_mode) {
    if (!info6) return fail(PGO_ERR_INVALID_ARG, "info_weighting = 1 needs the information matrices (pgo_create_weighted / pgo_create_from_graph)");
    if (opt.method == 2) return fail(PGO_ERR_UNSUPPORTED, "info_weighting is implemented for METHOD 0 and 1 only");
    for (int32_t e = 0; e < E; ++e) {  // every Omega must have a Cholesky factor
      const double* w = info6 + 6 * (size_t)e;
      const double l00 = w[0] > 0.0 ? std::sqrt(w[0]) : 0.0;
      const double l10 = l00 > 0.0 ? w[1] / l00 : 0.0, l20 = l00 > 0.0 ? w[2] / l00 : 0.0;
      const double d1 = w[3] - l10 * l10;
      const double l11 = d1 > 0.0 ? std::sqrt(d1) : 0.0;
      const double l21 = l11 > 0.0 ? (w[4] - l20 * l10) / l11 : 0.0;
      const double d2v = w[5] - l20 * l20 - l21 * l21;
      if (!(w[0] > 0.0) || !(d1 > 0.0) || !(d2v > 0.0) || !std::isfinite(d2v))
        return fail(PGO_ERR_NUMERIC, "info_weighting: the information matrix of edge " + std::to_string(e) +
                                         " is not positive definite (EDGE2 files are read positionally like EDGE_SE2, "
                                         "reference g2o_util.h:53-66)");
    }
    rec_doubles = dev::REC_INFO;
  }
  const char* fc = getenv("PGO_FORCE_COLLECTIVES");
  force_collectives = fc && fc[0] == '1';
  if (const char* nt = PGO_EXP_ENV("PGO_SPMV_NT")) spmv_nt = atoi(nt);
  if (const char* gc = getenv("PGO_GRAPH_COLLECTIVES")) graph_collectives = atoi(gc);
  grp_B = pgo::resolve_block_poses(opt.pcg_block_poses, N);
  chain_len = pgo::resolve_chain_len(opt.pcg_chain_len, opt.pcg_block_poses, N, E, ia, ib);
  if (chain_len != 0 && (chain_len < dev::CHAIN_CHUNK || chain_len % dev::CHAIN_CHUNK != 0 || dev::CHAIN_TILE % chain_len != 0))
    return fail(PGO_ERR_INVALID_ARG, "pcg_chain_len: a multiple of 4 that divides 256 (4 ... 256), 0 = off, -1 = auto");
  if (chain_len) grp_B = 1;
  // internal pose numbering
  // auto: several ranks (it shrinks every rank's halo 2.6-2.9x), and single-rank graphs too large for the direct solve, where
  // it is worth -11 % of K3's fabric traffic (977 -> 870 MB per product at 1M poses: the gathers of neighbouring tiles hit
  // the XCD's L2), -16 % of K2's and -22 % of K1's reads: 38.7 / 40.1 -> 39.7 / 41.5 GN it/s (profiles/r03_order.md)
  const bool reorder = opt.pose_ordering == 1 || (opt.pose_ordering < 0 && (world > 1 || N > DIRECT_MAX_POSES));
  std::vector<int32_t> ia_p, ib_p;
  std::vector<double> poses_p;
  fixed_internal = opt.fixed_pose;
  HIPC(hipSetDevice(device));
  HIPC(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  if (reorder) {
    // chain segments must stay contiguous under the renumbering.  The ordering is host work on the WHOLE edge list
    // (0.7 s at 1M poses): a process-wide cache serves repeated handles on the same graph, and with several ranks only
    // rank 0 computes it -- the others receive it through the communicator (a sum in which they contribute zeros: the one
    // collective both back-ends have for this), so a node does not spend ranks x 0.7 s of CPU on identical work.
    const int seg = std::max<int>(pgo::ORDER_SEGMENT, chain_len);
    if (world == 1) {
      PGOC(pgo::cached_pose_order(N, E, ia, ib, seg, &perm));
    } else {
      if (rank == 0) PGOC(pgo::cached_pose_order(N, E, ia, ib, seg, &perm));
      else perm.assign((size_t)N, 0);
      std::vector<double> tmp((size_t)N);
      for (int32_t i = 0; i < N; ++i) tmp[i] = (double)perm[i];
      double* d_tmp = nullptr;
      HIPC(hipMalloc((void**)&d_tmp, (size_t)N * sizeof(double)));
      HIPC(hipMemcpyAsync(d_tmp, tmp.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, stream));
      int st_b = PGO_OK;
      for (int64_t off = 0; off < N && st_b == PGO_OK; off += (1 << 16))   // 512 KiB pieces (a slot of the shm test back-end)
        if (comm->allreduce(d_tmp + off, (int)std::min<int64_t>(N - off, 1 << 16), false, stream) != 0)
          st_b = fail(PGO_ERR_COMM, "pose ordering: broadcast through the communicator failed");
      if (st_b == PGO_OK && hipMemcpyAsync(tmp.data(), d_tmp, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess)
        st_b = fail(PGO_ERR_HIP, "pose ordering: copy back");
      if (st_b == PGO_OK && hipStreamSynchronize(stream) != hipSuccess) st_b = fail(PGO_ERR_HIP, "pose ordering: synchronise");
      (void)hipFree(d_tmp);
      PGOC(st_b);
      for (int32_t i = 0; i < N; ++i) perm[i] = (int32_t)tmp[i];
    }
    ia_p.resize(E);
    ib_p.resize(E);
    for (int32_t e = 0; e < E; ++e) {
      ia_p[e] = perm[ia[e]];
      ib_p[e] = perm[ib[e]];
    }
    poses_p.resize((size_t)3 * N);
    for (int64_t i = 0; i < N; ++i) memcpy(&poses_p[(size_t)3 * perm[i]], poses_h + 3 * i, 3 * sizeof(double));
    ia = ia_p.data();
    ib = ib_p.data();
    poses_h = poses_p.data();
    if (fixed_internal >= 0) fixed_internal = perm[fixed_internal];
  }
  PGOC(pgo::build_shard_structure(N, E, ia, ib, meas, kind, opt.method, world, rank, chain_len ? chain_len : grp_B, &S,
                                  tile_breaks_h.empty() ? nullptr : &tile_breaks_h));
  n_full = (int64_t)world * S.rows_per_rank;
  const int64_t EL = S.n_edges_local, NL = S.n_loc;
  inc_stride = ((S.n_inc + 63) / 64) * 64;  // whole 64-incidence groups (dev::hoff_index)
  if (inc_stride == 0) inc_stride = 64;

  PGOC(dalloc(&poses, 3 * n_full));
  PGOC(dalloc(&cand, 3 * n_full));
  PGOC(dalloc(&scale, 3 * n_full));
  PGOC(dalloc(&p_full, dev::PS * n_full));
  PGOC(dalloc(&e_ia, EL));
  PGOC(dalloc(&e_ib, EL));
  PGOC(dalloc(&e_mx, EL));
  PGOC(dalloc(&e_my, EL));
  PGOC(dalloc(&e_mt, EL));
  PGOC(dalloc(&e_flags, EL));
  PGOC(dalloc(&jr, EL * rec_doubles));
  if (info6) PGOC(dalloc(&e_info, 6 * std::max<int64_t>(EL, 1)));
  PGOC(dalloc(&inc_ptr, NL + 1));
  PGOC(dalloc(&inc_edge, S.n_inc));
  PGOC(dalloc(&inc_col, S.n_inc));
  PGOC(dalloc(&tile_row, (int64_t)S.tile_row.size()));
  PGOC(dalloc(&inc_rowoff, std::max<int64_t>(S.n_inc, 1)));
  PGOC(dalloc(&hoff, 9 * inc_stride));
  PGOC(dalloc(&hd, 6 * NL));
  PGOC(dalloc(&gs, 3 * NL));
  PGOC(dalloc(&d2, 3 * NL));
  PGOC(dalloc(&minv, 6 * NL));
  PGOC(dalloc(&hdd, 3 * NL));
  PGOC(dalloc(&y, 3 * NL));
  PGOC(dalloc(&r, 3 * NL));
  PGOC(dalloc(&z, 3 * NL));
  PGOC(dalloc(&ap, 3 * NL));
  has_sw = (opt.method == 2);
  if (has_sw) {
    for (double** ptr : {&sw, &sw_cand, &sw_sigma, &sw_c, &sw_gamma, &sw_gs, &sw_den, &sw_hss}) PGOC(dalloc(ptr, EL));
    PGOC(dalloc(&sw_js, 3 * EL));
    PGOC(dalloc(&diag_full, 3 * NL));
    PGOC(dalloc(&gs_full, 3 * NL));
  }
  PGOC(dalloc(&st, 1));
  PGOC(dalloc(&scal, N_SCAL));
  PGOC(dalloc(&bad, 1));
  HIPC(hipHostMalloc((void**)&h_st, sizeof(dev::CgState)));
  HIPC(hipHostMalloc((void**)&h_scal, N_SCAL * sizeof(double)));

  auto cdiv = [](int64_t a, int64_t b) { return (int)((a + b - 1) / b); };
  auto up8 = [](int g) { return ((g + 7) / 8) * 8; };  // XCD-aware kernels need gridDim % 8 == 0
  g_edge = up8(std::max(1, cdiv(EL, dev::WG)));
  g_rows = std::max(1, cdiv(NL, dev::WG));
  g_vec = std::min(std::max(1, cdiv(NL, dev::WG)), 1024);
  g_flat = std::min(std::max(1, cdiv(3 * NL, dev::WG)), 1024);
  if (const char* fe = PGO_EXP_ENV("PGO_FLAT_GRID")) g_flat = std::min(g_flat, std::max(8, atoi(fe)));
  g_spmv = up8(std::min(std::max(1, S.n_tiles()), 2048));
  g_asm = up8(std::min(std::max(1, S.n_tiles()), 256 * 6));   // persistent workgroups: the pipelined K2 walks ~20 tiles each at 1M poses
  part_cap = std::max(g_edge, 2048) + 8 + 512;   // (+ the coarse level's dot partials behind the one-level r.z partials)
  for (int k = 0; k < N_PART; ++k) PGOC(dalloc(&part[k], part_cap));

  HIPC(hipMemcpyAsync(poses, poses_h, (size_t)3 * N * sizeof(double), hipMemcpyHostToDevice, stream));
  PGOC(sync());  // poses_p (reordered copy) dies with this call
  PGOC(upload(e_ia, S.ia));
  PGOC(upload(e_ib, S.ib));
  PGOC(upload(e_mx, S.mx));
  PGOC(upload(e_my, S.my));
  PGOC(upload(e_mt, S.mt));
  PGOC(upload(e_flags, S.flags));
  if (info6) {  // planes over the local edges
    std::vector<double> planes((size_t)6 * EL);
    for (int64_t k = 0; k < EL; ++k)
      for (int c = 0; c < 6; ++c) planes[(size_t)c * EL + k] = info6[6 * (size_t)S.orig_edge[k] + c];
    PGOC(upload(e_info, planes));
    PGOC(sync());  // `planes` dies with this scope
  }
  PGOC(upload(inc_ptr, S.inc_ptr));
  PGOC(upload(inc_edge, S.inc_edge));
  PGOC(upload(inc_col, S.inc_col));
  PGOC(upload(tile_row, S.tile_row));
  PGOC(upload(inc_rowoff, S.inc_rowoff));
  {  // one 16-byte descriptor per tile for K3: {first local row, rows, first incidence, incidences}
    // in breadth-first order of the tile graph (compute_tile_order): tiles running together gather the same lines
    std::vector<int4> desc((size_t)std::max(1, S.n_tiles()));
    std::vector<int32_t> order;
    const char* to = PGO_EXP_ENV("PGO_TILE_ORDER");
    // OFF by default: it cuts the gather traffic (FETCH_SIZE 992 -> 920-937 MB at 1M poses) but the scattered 18-KB
    // H chunks cost more than that saves (184 vs 179 us); PGO_TILE_ORDER=1 turns it on (never in a batch, which keeps
    // each problem's tiles together)
    if (to && to[0] == '1' && S.n_tiles() >= 4096 && !batch_mode) pgo::compute_tile_order(S, &order);
    for (int k = 0; k < S.n_tiles(); ++k) {
      const int t = order.empty() ? k : order[k];
      const int32_t r0 = S.tile_row[t], r1 = S.tile_row[t + 1];
      desc[k] = make_int4(r0, r1 - r0, S.inc_ptr[r0], S.inc_ptr[r1] - S.inc_ptr[r0]);
    }
    {
      // the software-pipelined product kernel (k_spmv_p) needs plain tiles: no chunked heavy row, at most 85 rows
      // (one row-phase pass); PGO_SPMV_PIPE=0 keeps k_spmv_t.  Measured on one box at 1M poses: k_spmv_t 185.7 us (8
      // workgroups per CU), k_spmv_p 172.4 / 175.5 / 168.8 / 165.1 us at 8 / 6 / 5 / 4 workgroups per CU.
      bool ok = knob("spmv_pipe") != 0;   // (test hook: 0 keeps k_spmv_t so that the two product kernels can be compared)
      for (int t = 0; ok && t < S.n_tiles(); ++t)
        ok = desc[t].w <= dev::WG && desc[t].y * 3 <= dev::WG;
      spmv_pipe = ok;
      if (ok) {
        int per_cu = 4;
        if (const char* ge = PGO_EXP_ENV("PGO_SPMV_PIPE_WGS")) per_cu = std::max(1, atoi(ge));
        g_spmv = ((std::min(std::max(1, S.n_tiles()), 256 * per_cu) + 7) / 8) * 8;
      }
    }
    PGOC(dalloc(&tile_desc, (int64_t)desc.size()));
    PGOC(upload(tile_desc, desc));
    PGOC(sync());  // `desc` dies with this scope
  }
  if (has_sw) {  // switches start at 1.0 (main.cpp:117,139)
    std::vector<double> ones((size_t)EL, 1.0);
    PGOC(upload(sw, ones));
    PGOC(upload(sw_cand, ones));
    PGOC(sync());  // `ones` dies with this scope
  }
  // halo lists for the point-to-point exchange of the search direction
  use_halo = world > 1 && opt.halo_exchange != 0;
  if (use_halo) {
    PGOC(dalloc(&halo_send_rows, (int64_t)S.halo_send_row.size()));
    PGOC(dalloc(&halo_recv_rows, (int64_t)S.halo_recv_row.size()));
    PGOC(dalloc(&halo_send_buf, 3 * (int64_t)S.halo_send_row.size()));
    PGOC(dalloc(&halo_recv_buf, 3 * (int64_t)S.halo_recv_row.size()));
    PGOC(upload(halo_send_rows, S.halo_send_row));
    PGOC(upload(halo_recv_rows, S.halo_recv_row));
    halo_send_off3.resize(S.halo_send_off.size());
    halo_recv_off3.resize(S.halo_recv_off.size());
    for (size_t k = 0; k < S.halo_send_off.size(); ++k) {
      halo_send_off3[k] = 3 * S.halo_send_off[k];
      halo_recv_off3[k] = 3 * S.halo_recv_off[k];
    }
    if (opt.halo_overlap != 0 && S.n_tiles() > 0) {
      std::vector<int32_t> rows, ptr(1, 0), slots;
      for (int32_t r = 0; r < S.n_loc; ++r) {
        const size_t before = slots.size();
        for (int32_t q = S.inc_ptr[r]; q < S.inc_ptr[r + 1]; ++q)
          if (S.inc_col[q] < S.lo || S.inc_col[q] >= S.hi) slots.push_back(q);
        if (slots.size() > before) {
          rows.push_back(r);
          ptr.push_back((int32_t)slots.size());
        }
      }
      n_rr = (int)rows.size();
      PGOC(dalloc(&rr_rows, std::max<int64_t>(1, n_rr)));
      PGOC(dalloc(&rr_ptr, (int64_t)ptr.size()));
      PGOC(dalloc(&rr_slots, std::max<int64_t>(1, (int64_t)slots.size())));
      PGOC(upload(rr_rows, rows));
      PGOC(upload(rr_ptr, ptr));
      PGOC(upload(rr_slots, slots));
      PGOC(sync());  // the lists die with this scope
      g_spmv_loc = up8(std::min(std::max(1, S.n_tiles()), 1536));  // + g_rr <= 2048 partials
      g_rr = std::min(std::max(1, (n_rr + dev::WG - 1) / dev::WG), 512);
      HIPC(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
      HIPC(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming));
      HIPC(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming));
      overlap = true;
    }
  }
  // preconditioner block size
  if (grp_B > 1 && NL > 0) {
    grp_nb = 3 * grp_B;
    grp_pad = grp_nb;  // lanes per group in the apply kernels (any value <= WG works: slot = tid / grp_pad)
    n_groups = (int)((NL + grp_B - 1) / grp_B);
    const int gpw = dev::WG / grp_pad;
    g_grp = std::min(std::max(1, (n_groups + gpw - 1) / gpw), 2048);
    const int prep_gpw = (grp_nb <= 24) ? dev::WG / 64 : 1;  // groups per workgroup in k_prepare_groups
    grp_lds = (size_t)prep_gpw * grp_nb * (grp_nb + 1) * sizeof(double);
    grp_prep_grid = std::min((n_groups + prep_gpw - 1) / prep_gpw, 65536);
    PGOC(dalloc(&ginv, (int64_t)n_groups * grp_nb * grp_nb));
    if (grp_lds > 48 * 1024)
      HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_prepare_groups), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)grp_lds));
  } else {
    grp_B = 1;
  }
  if (chain_len && NL > 0) {
    chain_pad = (int)(((NL + dev::CHAIN_TILE - 1) / dev::CHAIN_TILE) * dev::CHAIN_TILE);
    PGOC(dalloc(&chain_c, (int64_t)dev::CHAIN_REC * NL));   // zero-filled: rows without a block (i, i-1) keep C = 0
    {
      std::vector<int32_t> dup;
      for (int32_t r = 1; r < S.n_loc; ++r) {
        if ((r % chain_len) == 0) continue;
        int cnt = 0;
        for (int32_t q = S.inc_ptr[r]; q < S.inc_ptr[r + 1]; ++q) cnt += (S.inc_col[q] == S.lo + r - 1);
        if (cnt > 1) dup.push_back(r);
      }
      n_chain_dup = (int)dup.size();
      if (n_chain_dup) {
        PGOC(dalloc(&chain_dup_rows, n_chain_dup));
        PGOC(upload(chain_dup_rows, dup));
        PGOC(sync());  // `dup` dies with this scope
      }
    }
    PGOC(dalloc(&chain_w, 9 * (int64_t)chain_pad));
    PGOC(dalloc(&chain_s, 6 * (int64_t)chain_pad));
    // the padding rows of the factor planes are never written: they must read as 0
    HIPC(hipMemsetAsync(chain_w, 0, (size_t)9 * chain_pad * sizeof(double), stream));
    HIPC(hipMemsetAsync(chain_s, 0, (size_t)6 * chain_pad * sizeof(double), stream));
    g_chain = (int)std::min<int64_t>((NL + 4 * dev::CHAIN_TILE - 1) / (4 * dev::CHAIN_TILE), 2048);
    // apply kernel: the lean form (one DPP-shift recurrence step per lane of a segment), 2 poses per lane for segments of
    // <= 64 poses, 4 for longer ones (INTEL, chain-256: 20 us per apply in the scan form -- five 256-row tiles, latency-
    // bound -- of a 30 us PCG iteration); PGO_CHAIN_KERNEL = scan | lean2 | lean4 overrides (experiments)
    chain_chunk = chain_len <= 64 ? 2 : 4;   // 4: segments of up to 256 poses (the small chain-like graphs)
    if (const char* ck = PGO_EXP_ENV("PGO_CHAIN_KERNEL")) {
      if (!strcmp(ck, "scan")) chain_chunk = 0;
      else if (!strcmp(ck, "lean2") && (128 % chain_len) == 0) chain_chunk = 2;
      else if (!strcmp(ck, "lean4")) chain_chunk = 4;
    }
    if (chain_chunk) {
      chain_steps = chain_len / chain_chunk - 1;
      const int64_t n_wt = (NL + 64 * chain_chunk - 1) / (64 * chain_chunk);
      // Small graphs (few tiles, nothing to overlap with): one wavefront per workgroup, so that every tile loads through
      // its own CU's L1, and -- for segments of more than 16 lanes -- the recurrence as a log-depth scan (INTEL, 256-pose
      // segments: 7.4 -> 4.8 us per apply); large graphs keep the 4-wave workgroups and the serial DPP recurrence,
      // which needs fewer registers and no LDS-crossbar shuffles.
      int64_t small_max = 512;
      if (const char* sm = PGO_EXP_ENV("PGO_CHAIN_SMALL_TILES")) small_max = atoll(sm);
      const bool small = n_wt <= small_max;
      chain_nw = small ? 1 : 4;
      chain_scan = 0;
      if (small && chain_len / chain_chunk > 16)
        for (int l = 1; l < chain_len / chain_chunk; l <<= 1) ++chain_scan;
      if (const char* cs = PGO_EXP_ENV("PGO_CHAIN_SCAN")) {  // experiments: 0 = always serial
        if (atoi(cs) == 0) chain_scan = 0;
      }
      // every workgroup of the NEXT kernel re-sums this kernel's per-workgroup partials, so fewer, longer-running
      // workgroups are cheaper all round: 2048 -> 512 (and 1024 for the flat vector kernels) 25.43 -> 24.65 ms per LM
      // iteration at 1M poses (same box, 3 interleaved repetitions)
      int cap = 512;
      if (const char* ce = PGO_EXP_ENV("PGO_CHAIN_GRID")) cap = std::max(8, atoi(ce));
      g_chain = (int)std::min<int64_t>((n_wt + chain_nw - 1) / chain_nw, chain_nw == 1 ? 2048 : cap);
    }
  } else {
    chain_len = 0;
  }
  // One-workgroup PCG (solo.hip.h): a single rank, a chain or 3x3 block-Jacobi preconditioner, no chunked heavy row.
  {
    const char* se = PGO_EXP_ENV("PGO_SOLO");
    // a single graph takes this path only on request (PGO_SOLO=1): one CU's L1 paces the solve -- INTEL 36 us per PCG
    // iteration against 27 us for the three-kernel loop on 256 CUs, MIT / FR079 8 % faster -- the win is the BATCH, where
    // every problem has a CU of its own
    bool ok = world == 1 && !force_collectives && grp_B == 1 && NL > 0 &&
              (batch_mode || ((se && se[0] == '1') && S.n_tiles() <= 64));
    for (int t = 0; ok && t < S.n_tiles(); ++t)
      ok = S.inc_ptr[S.tile_row[t + 1]] - S.inc_ptr[S.tile_row[t]] <= dev::WG;
    if (ok && chain_len) {
      chain_chunk = dev::SOLO_CH;  // the 256-row tile layout of the factor planes
      chain_steps = chain_len / chain_chunk - 1;
      const int64_t n_wt = (NL + 64 * chain_chunk - 1) / (64 * chain_chunk);
      chain_nw = n_wt <= 512 ? 1 : 4;
      chain_scan = 0;
      g_chain = (int)std::min<int64_t>((n_wt + chain_nw - 1) / chain_nw, 2048);
      const int lanes = chain_len / dev::SOLO_CH;  // lanes per segment: serial recurrence up to 16, else log-depth scan
      solo_steps = lanes - 1;
      solo_scan = 0;
      if (lanes > 16)
        for (int l = 1; l < lanes; l <<= 1) ++solo_scan;
    }
    if (batch_mode && !ok) return fail(PGO_ERR_UNSUPPORTED, "pgo_batch: a problem has a row with more than 256 incidences");
    if (ok && !batch_mode) {
      PGOC(dalloc(&solo_prob, 1));
      PGOC(dalloc(&solo_out, 1));
      HIPC(hipHostMalloc((void**)&h_solo, sizeof(dev::SoloOut)));
    }
    solo = ok;
  }
  {
    int64_t fused_max = 16384;
    if (const char* fm = PGO_EXP_ENV("PGO_FUSED_MAX_ROWS")) fused_max = atoll(fm);
    fused_p = knob("fused_p") != 0 && world == 1 && !force_collectives && !batch_mode && NL > 0 && NL <= fused_max;
    if (fused_p) PGOC(dalloc(&p_full2, dev::PS * n_full));
  }
  {
    // One reduction point per PCG iteration instead of two (k_cg_sr_*): where the all-reduces are latency -- several ranks --
    // and only in the inexact mode (pcg_rtol >= 1e-6: the recurrence for A p drifts over the thousands of iterations of
    // the exact mode); needs the lean chain apply.  Test hook "single_reduction": 1 = also on one rank, 0 = never.
    const long long kn = knob("single_reduction");
    use_sr = chain_len > 0 && chain_chunk > 0 && !solo && !fused_p && !batch_mode && NL > 0 && opt.pcg_rtol >= 1e-6 &&
             (kn == 1 || (kn != 0 && (world > 1 || force_collectives)));
    if (use_sr) PGOC(dalloc(&sr_s, 3 * NL));
  }
  if (!fixed_mask_h.empty()) {
    PGOC(dalloc(&fixed_mask, (int64_t)fixed_mask_h.size()));
    PGOC(upload(fixed_mask, fixed_mask_h));
  }
  if (batch_mode) PGOC(dalloc(&edge_cost, std::max<int64_t>(EL, 1)));
  PGOC(direct_setup(N));
  PGOC(coarse_setup());   // (after the direct solver's decision: auto adds the coarse level only to solves that stay on PCG)
  return sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Second preconditioner level (coarse.hip.h).  opt.pcg_coarse_poses: 0 = off, > 0 = poses per aggregate, -1 = auto.
namespace {
constexpr int COARSE_MAX_RANK = 6144;   // order of the dense coarse matrix (k_chol_panel's range)
constexpr int COARSE_EXPLICIT_RANK = 1024;   // up to here the explicit inverse N'N is formed once per LM iteration
}
int pgo_handle::coarse_setup() {
  const int world = comm ? comm->world : 1;
  const int64_t NL = S.n_loc;
  int want = opt.pcg_coarse_poses;
  if (want == 0) return PGO_OK;
  auto no = [&](const std::string& why) -> int {
    if (want > 0) return fail(PGO_ERR_UNSUPPORTED, "pcg_coarse_poses: " + why);
    return PGO_OK;
  };
  if (world != 1 || force_collectives) return no("one rank only");
  if (batch_mode) return no("not inside a batched handle");
  if (NL < 2) return PGO_OK;
  if (want < 0) {
    // auto: the exact mode of graphs that stay on PCG (the direct solve, where it applies cheaply, is faster still).
    // Aggregates of 16 poses up to ~10k poses (M3500: rank 657), growing so that the coarse matrix stays below rank ~2400
    // (its Cholesky is paid once per LM iteration, its two dense products once per PCG iteration)
    if (!(opt.pcg_rtol <= 1e-4) || direct || NL < 512) return PGO_OK;
    want = 16;
    while (3 * ((NL + want - 1) / want) > 2400) want *= 2;
  }
  co_agg = want;
  co_nagg = (int)((NL + co_agg - 1) / co_agg);
  co_K = 3 * co_nagg;
  if (co_K + 1 > COARSE_MAX_RANK) return no("the coarse matrix would have order " + std::to_string(co_K) + " (at most " + std::to_string(COARSE_MAX_RANK - 1) + ")");
  co_Kp = ((co_K + dev::CHOL_NB - 1) / dev::CHOL_NB) * dev::CHOL_NB;
  // coarse blocks and their fine entries: incidences sorted by (aggregate of the row, aggregate of the column, position)
  std::vector<uint64_t> key((size_t)S.n_inc);
  {
    size_t k = 0;
    for (int32_t r = 0; r < S.n_loc; ++r)
      for (int32_t q = S.inc_ptr[r]; q < S.inc_ptr[r + 1]; ++q) {
        const uint64_t I = (uint64_t)(r / co_agg), J = (uint64_t)((S.inc_col[q] - S.lo) / co_agg);
        key[k++] = ((I * (uint64_t)co_nagg + J) << 32) | (uint32_t)q;
      }
  }
  std::sort(key.begin(), key.end());
  std::vector<int32_t> row_of((size_t)S.n_inc);
  for (int32_t r = 0; r < S.n_loc; ++r)
    for (int32_t q = S.inc_ptr[r]; q < S.inc_ptr[r + 1]; ++q) row_of[q] = r;
  std::vector<int32_t> cbi, cbj, cbp, cbq((size_t)S.n_inc), cbr((size_t)S.n_inc);
  {
    size_t k = 0;
    int next_diag = 0;   // every aggregate gets its (I, I) block, also without an off-diagonal fine entry inside
    auto open_block = [&](int I, int J) {
      cbi.push_back(I);
      cbj.push_back(J);
      cbp.push_back((int32_t)k);
    };
    while (k < key.size() || next_diag < co_nagg) {
      const uint64_t blk = k < key.size() ? (key[k] >> 32) : ~0ull;
      const uint64_t dblk = next_diag < co_nagg ? (uint64_t)next_diag * co_nagg + next_diag : ~0ull;
      if (dblk < blk) {   // a diagonal block without fine off-diagonal entries
        open_block(next_diag, next_diag);
        ++next_diag;
        continue;
      }
      if (dblk == blk) ++next_diag;
      open_block((int)(blk / co_nagg), (int)(blk % co_nagg));
      while (k < key.size() && (key[k] >> 32) == blk) {
        const int32_t q = (int32_t)(key[k] & 0xffffffffu);
        cbq[k] = q;
        cbr[k] = row_of[q];
        ++k;
      }
    }
    cbp.push_back((int32_t)k);
  }
  co_ncb = (int)cbi.size();
  PGOC(dalloc(&co_pb, 5 * NL));
  PGOC(dalloc(&co_cap, (int64_t)co_Kp * co_Kp));
  PGOC(dalloc(&co_nm, (int64_t)co_Kp * co_Kp));
  PGOC(dalloc(&co_dwork, (int64_t)(co_Kp / 32) * 1024));
  PGOC(dalloc(&co_rc, co_Kp));
  PGOC(dalloc(&co_cy, co_Kp));
  PGOC(dalloc(&co_ec, co_Kp));
  PGOC(dalloc(&co_ok, 1));
  if (co_Kp <= COARSE_EXPLICIT_RANK) PGOC(dalloc(&co_ainv, (int64_t)co_Kp * co_Kp));
  co_ndot = co_ainv ? (co_Kp + 3) / 4 : (co_Kp + 255) / 256;
  if (g_chain + co_ndot + 8 > part_cap || g_grp + co_ndot + 8 > part_cap || g_vec + co_ndot + 8 > part_cap)
    return no("internal: not enough room for the coarse level's dot partials");
  PGOC(dalloc(&co_cb_i, co_ncb));
  PGOC(dalloc(&co_cb_j, co_ncb));
  PGOC(dalloc(&co_cb_ptr, co_ncb + 1));
  PGOC(dalloc(&co_cb_q, std::max<int64_t>(1, S.n_inc)));
  PGOC(dalloc(&co_cb_row, std::max<int64_t>(1, S.n_inc)));
  PGOC(upload(co_cb_i, cbi));
  PGOC(upload(co_cb_j, cbj));
  PGOC(upload(co_cb_ptr, cbp));
  PGOC(upload(co_cb_q, cbq));
  PGOC(upload(co_cb_row, cbr));
  PGOC(sync());   // the host lists die with this scope
  HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_chol_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::CHOL_LDS_BYTES));
  use_coarse = true;
  if (opt.linear_solver == 0) dl_possible = false;   // (ranks above the direct solve's cheap range: two-level PCG instead of the PCG / direct alternation)
  // the loops that fold launches together assume the one-level preconditioner: two-level solves take the plain three-kernel loop
  solo = false;
  fused_p = false;
  use_sr = false;
  return PGO_OK;
}

int pgo_handle::coarse_factor() {
  dev::CoarseArgs A;
  A.n_loc = S.n_loc;
  A.agg = co_agg;
  A.n_agg = co_nagg;
  A.K = co_K;
  A.Kp = co_Kp;
  A.poses = poses;
  A.scale = scale;
  A.pb = co_pb;
  A.hoff = hoff;
  A.hd = hd;
  A.d2 = d2;
  A.inc_col = inc_col;
  A.cb_i = co_cb_i;
  A.cb_j = co_cb_j;
  A.cb_ptr = co_cb_ptr;
  A.cb_q = co_cb_q;
  A.cb_row = co_cb_row;
  A.n_cb = co_ncb;
  A.cap = co_cap;
  A.dwork = co_dwork;
  hipLaunchKernelGGL(dev::k_coarse_basis, dim3((co_nagg + 3) / 4), dim3(256), 0, stream, A);
  PGOC(check_launch("k_coarse_basis"));
  HIPC(hipMemsetAsync(co_cap, 0, (size_t)co_Kp * co_Kp * sizeof(double), stream));
  HIPC(hipMemsetAsync(co_dwork, 0, (size_t)(co_Kp / 32) * 1024 * sizeof(double), stream));
  hipLaunchKernelGGL(dev::k_coarse_assemble, dim3((co_ncb + 3) / 4), dim3(256), 0, stream, A);
  PGOC(check_launch("k_coarse_assemble"));
  if (co_Kp > co_K) {
    hipLaunchKernelGGL(dev::k_coarse_pad, dim3(1), dim3(32), 0, stream, co_cap, co_dwork, co_K, co_Kp);
    PGOC(check_launch("k_coarse_pad"));
  }
  const int nb = co_Kp / 32;
  for (int kb = 0; kb < nb; ++kb) {
    hipLaunchKernelGGL(dev::k_chol_panel, dim3(std::max(1, nb - 1)), dim3(dev::CHOL_THREADS), dev::CHOL_LDS_BYTES, stream, co_cap, co_nm, co_dwork, co_Kp, nb, kb);
    PGOC(check_launch("k_chol_panel (coarse level)"));
  }
  // usable?  a probe through the factor: x = A_c^-1 1 must be finite (a pivot lost to rounding leaves NaNs behind it)
  hipLaunchKernelGGL(dev::k_fill, dim3((co_Kp + 255) / 256), dim3(256), 0, stream, co_rc, (int64_t)co_Kp, 1.0);
  if (co_ainv) {
    hipLaunchKernelGGL(dev::k_coarse_ainv, dim3((co_Kp + 255) / 256, co_Kp), dim3(256), 0, stream, (const double*)co_nm, co_Kp, co_ainv);
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)co_ainv, co_Kp, nb, (const double*)co_rc, co_ec, 0);
  } else {
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_rc, co_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_cy, co_ec, 1);
  }
  hipLaunchKernelGGL(dev::k_coarse_check, dim3(1), dim3(256), 0, stream, (const double*)co_ec, co_Kp, co_ok);
  return check_launch("coarse level probe");
}

int pgo_handle::coarse_solve(double* dot_part, const int32_t* done) {
  const int nb = co_Kp / 32;
  hipLaunchKernelGGL(dev::k_coarse_restrict, dim3((co_nagg + 3) / 4), dim3(256), 0, stream, (int)S.n_loc, co_agg, co_nagg, (const double*)co_pb,
                     (const double*)r, co_rc, done);
  if (co_ainv) {
    hipLaunchKernelGGL(dev::k_coarse_matvec, dim3(co_ndot), dim3(256), 0, stream, (const double*)co_ainv, co_Kp, (const double*)co_rc, co_ec,
                       dot_part, (const int32_t*)co_ok, done);
  } else {
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_rc, co_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_cy, co_ec, 1);
    hipLaunchKernelGGL(dev::k_coarse_dot, dim3(co_ndot), dim3(256), 0, stream, co_Kp, (const double*)co_rc, co_ec, dot_part,
                       (const int32_t*)co_ok, done);
  }
  return check_launch("coarse level solve");
}

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
  hipLaunchKernelGGL(dev::k_switch_prepare, dim3(g_sw), dim3(dev::WG), 0, stream, switch_arrays(), (const double*)jr, radius,
                     opt.min_lm_diagonal, opt.max_lm_diagonal, part[2], part[3]);
  PGOC(check_launch("k_switch_prepare"));
  PGOC(assemble_enqueue());
  hipLaunchKernelGGL(dev::k_grad_max, dim3(g_flat), dim3(dev::WG), 0, stream, (const double*)gs_full, (const double*)scale, S.n_loc,
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
  hipLaunchKernelGGL(dev::k_jacobi_scale, dim3(g_rows), dim3(dev::WG), 0, stream, hd, S.n_loc, S.lo, fixed, 0, scale, (const uint8_t*)fixed_mask);
  PGOC(check_launch("k_jacobi_scale"));
  PGOC(allgather(scale));
  PGOC(linearize(false));
  const double cost0 = h_scal[0];
  if (has_sw) {  // Jacobi scale of the switch columns from the iteration-0 Jacobian
    hipLaunchKernelGGL(dev::k_switch_scale, dim3((S.n_edges_local + 255) / 256 + 1), dim3(256), 0, stream, switch_arrays(),
                       opt.jacobi_scaling);
    PGOC(check_launch("k_switch_scale"));
  }
  if (opt.jacobi_scaling) {
    hipLaunchKernelGGL(dev::k_jacobi_scale, dim3(g_rows), dim3(dev::WG), 0, stream, hd, S.n_loc, S.lo, fixed, 1, scale, (const uint8_t*)fixed_mask);
    PGOC(check_launch("k_jacobi_scale"));
    PGOC(allgather(scale));
    PGOC(linearize(true));  // the records do not depend on the scales: re-assemble only
  }
  cost = initial_cost = cost0;
  hipLaunchKernelGGL(dev::k_grad_max, dim3(g_flat), dim3(dev::WG), 0, stream, gs, scale, S.n_loc, S.lo, part[0]);
  PGOC(check_launch("k_grad_max"));
  PGOC(reduce_to_scal({{part[0], g_flat, 1}}, 2, true));
  hipLaunchKernelGGL(dev::k_xnorm, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, poses, scale, part[1]);
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

// ---------------------------------------------------------------------------------------------------------------
// Direct solve for small chain-like graphs (direct.hip.h).  opt.linear_solver: 0 = auto, 1 = PCG, 2 = direct.
// Auto picks it in the "exact" mode only (pcg_rtol <= 1e-8, where PCG stands in for the reference's
// SPARSE_NORMAL_CHOLESKY, main.cpp:154-163) and only while the caller left the preconditioner to the library; it needs
// one rank, METHOD 0 / 1, a constant pose, an edge between every pair of consecutive poses, and few enough other edges.
namespace {
constexpr int DIRECT_MAX_RANK = 6144;   // 3 x (edges outside the chain) + 1: order of the dense capacitance matrix
constexpr double PCG_SECONDS_PER_ITER_SMALL = 14e-6;   // a PCG iteration of the two-launch loop on graphs of a few thousand poses
constexpr int DIRECT_PROBE_EVERY = 10;
constexpr int DIRECT_AUTO_RANK = 2048;  // auto takes the direct solve up to this rank (INTEL + 50: 918 -> 1.5 ms per LM iteration; FRH, 4515: 13 ms
                                        // against 29 ms of PCG, but one refinement step leaves 5e-8 there; M3500, 5862: 22 ms, the same as PCG)
}  // namespace

int pgo_handle::direct_setup(int32_t N, bool switch_now) {
  const int world = comm ? comm->world : 1;
  int want = switch_now ? 2 : opt.linear_solver;
  if (const char* de = PGO_EXP_ENV("PGO_DIRECT")) want = atoi(de) ? 2 : 1;  // experiment builds only: force on / off
  if (want == 1) return PGO_OK;
  if (want != 0 && want != 2) return fail(PGO_ERR_INVALID_ARG, "linear_solver: 0 = auto, 1 = PCG, 2 = direct (chain + low rank)");
  auto no = [&](const std::string& why) -> int {
    if (want == 2) return fail(PGO_ERR_UNSUPPORTED, "linear_solver = direct: " + why);
    return PGO_OK;
  };
  if (want == 0 && (!(opt.pcg_rtol <= 1e-8) || opt.pcg_chain_len != -1 || opt.pcg_block_poses != 0)) return PGO_OK;
  if (world != 1 || force_collectives) return no("one rank only");
  if (batch_mode) return no("not inside a batched handle");
  // information weighting: the chain blocks inherit the information matrices' condition numbers (INTEL: 1e11) and the
  // Woodbury correction loses the solution (measured: residual 1e-3 after refinement); MIT-like inputs would work, but the
  // library cannot tell from the graph -- PCG there
  if (info_mode) return no("not with information weighting");
  if (fixed_internal < 0) return no("needs a constant pose (it anchors the chain)");
  if (!perm.empty()) return no("the internal pose ordering is on");
  if (N < 2 || N > DIRECT_MAX_POSES) return no("2 .. " + std::to_string(DIRECT_MAX_POSES) + " poses");
  const int64_t EL = S.n_edges_local;
  std::vector<int32_t> chain((size_t)N, -1), lr, va, vb;
  for (int64_t e = 0; e < EL; ++e) {
    const int32_t lo_p = std::min(S.ia[e], S.ib[e]), hi_p = std::max(S.ia[e], S.ib[e]);
    if (hi_p == lo_p + 1 && chain[lo_p] < 0) {
      chain[lo_p] = (int32_t)e;
    } else {
      lr.push_back((int32_t)e);
      va.push_back(S.ia[e]);
      vb.push_back(S.ib[e]);
    }
  }
  for (int32_t i = 0; i + 1 < N; ++i)
    if (chain[i] < 0) return no("poses " + std::to_string(i) + " and " + std::to_string(i + 1) + " are not joined by an edge");
  dl_m = (int)lr.size();
  dl_K = 3 * dl_m;
  if (want == 0 && dl_K > DIRECT_AUTO_RANK) {
    // beyond this rank the dense Cholesky is no longer cheap (M3500, 5862: 12.7 ms per LM iteration) and whether PCG beats it
    // depends on the conditioning, which nobody knows beforehand (M3500 with DCS: 1460 PCG iterations per LM iteration =
    // 21 ms; without: 230 = 4 ms).  So the handle starts with PCG and lm_iteration() switches to the direct solve once a
    // PCG solve has cost more than the direct one would (both from counts, not clocks: the decision is reproducible)
    if (dl_K + 1 <= DIRECT_MAX_RANK) {
      dl_possible = true;
      // cost model of a direct solve: the capacitance Cholesky (M3500, rank 5862: 11.7 ms; ~ rank^2.5 between INTEL, FRH and
      // M3500) + the chain (factorisation and sweeps: 0.15 us per pose, 5k .. 40k-pose chains) + 1 ms of fixed latencies
      const double kk = dl_K / 5862.0;
      dl_est_seconds = 1.0e-3 + 11.7e-3 * kk * kk * std::sqrt(kk) + 0.15e-6 * N;
    }
    return PGO_OK;
  }
  if (dl_K + 1 > DIRECT_MAX_RANK) return no(std::to_string(dl_m) + " edges outside the odometry chain (at most " + std::to_string((DIRECT_MAX_RANK - 1) / 3) + ")");
  dl_Kp = std::max(dev::CHOL_NB, ((dl_K + dev::CHOL_NB - 1) / dev::CHOL_NB) * dev::CHOL_NB);
  // separators: the chain is factorised in nsep + 1 pieces side by side (k_dlr_factor is one wavefront's dependent chain:
  // 0.3 us per pose); PGO_DIRECT_SEP=0 keeps one piece
  dl_nsep = 0;
  {
    const char* se = PGO_EXP_ENV("PGO_DIRECT_SEP");
    if (N >= 256 && !(se && se[0] == '0')) {
      // 3 separators up to ~5000 poses (INTEL / MIT: 3 -> 826 / 2392 GN it/s, 5 -> 813 / 2294, 7 -> 788 / 2116, 15 -> 606 / 1209:
      // every separator adds 3 columns and a row of the Schur system), then one per ~1200 poses (40k poses: 3 -> 156, 15 -> 205)
      dl_nsep = std::min(dev::DLR_MAX_SEP, std::max(3, (int)(N / 1200)));
      if (const char* ne = PGO_EXP_ENV("PGO_DIRECT_NSEP")) dl_nsep = std::min(dev::DLR_MAX_SEP, std::max(1, atoi(ne)));   // experiments
      for (int j = 0; j < dl_nsep; ++j) dl_sep[j] = (int)(((int64_t)(j + 1) * N) / (dl_nsep + 1));
    }
  }
  dl_nU = 3 * dl_nsep;
  dl_ld = ((dl_K + 1 + dl_nU + 63) / 64) * 64;
  dl_refine = 1;   // (a second step does not lower FRH's 5e-8: that residual is what the conditioning allows)
  if (const char* re = PGO_EXP_ENV("PGO_DIRECT_REFINE")) dl_refine = std::max(0, atoi(re));
  if (knob("direct_fail_at") > 0) dl_fail_at = (int)knob("direct_fail_at");   // test hook
  if (const char* ge = PGO_EXP_ENV("PGO_DIRECT_GRAPH")) dl_use_graph = ge[0] == '1';
  PGOC(dalloc(&dl_chain_edge, N));
  PGOC(dalloc(&dl_lr_edge, std::max(1, dl_m)));
  PGOC(dalloc(&dl_va, std::max(1, dl_m)));
  PGOC(dalloc(&dl_vb, std::max(1, dl_m)));
  PGOC(upload(dl_chain_edge, chain));
  PGOC(upload(dl_lr_edge, lr));
  PGOC(upload(dl_va, va));
  PGOC(upload(dl_vb, vb));
  if (knob("direct_setup_fail") > 0) return fail(PGO_ERR_NOMEM, "hipMalloc: out of memory (test hook direct_setup_fail)");
  PGOC(dalloc(&dl_trec, (int64_t)dev::DLR_REC * N));
  PGOC(dalloc(&dl_fac, (int64_t)dev::DLR_REC * (N + 1)));
  PGOC(dalloc(&dl_vrec, (int64_t)dev::DLR_V * std::max(1, dl_m)));
  PGOC(dalloc(&dl_Z, (int64_t)3 * N * dl_ld));
  PGOC(dalloc(&dl_x1, (int64_t)3 * N * 64));
  PGOC(dalloc(&dl_cap, (int64_t)dl_Kp * dl_Kp));
  PGOC(dalloc(&dl_dwork, (int64_t)(dl_Kp / 32) * 1024));
  PGOC(dalloc(&dl_nm, (int64_t)dl_Kp * dl_Kp));
  PGOC(dalloc(&dl_cy, dl_Kp));
  PGOC(dalloc(&dl_pre, (int64_t)dev::DLR_PRE * N));
  int want_seg = 32;   // segments the chain sweeps are cut into (PGO_DIRECT_NSEG: experiments, <= 64)
  if (const char* ns = PGO_EXP_ENV("PGO_DIRECT_NSEG")) want_seg = std::min(dev::DLR_MAX_SEG, std::max(1, atoi(ns)));
  dl_seglen = std::max(1, (N + want_seg - 1) / want_seg);
  dl_nseg = (N + dl_seglen - 1) / dl_seglen;
  if (N <= 4096 && !PGO_EXP_ENV("PGO_DIRECT_NO_SOLVE1")) {
    dl_seglen2 = std::max(1, (N + 255) / 256);
    dl_nseg2 = (N + dl_seglen2 - 1) / dl_seglen2;
    PGOC(dalloc(&dl_pre2, (int64_t)dev::DLR_PRE * N));
  }
  PGOC(dalloc(&dl_E, (int64_t)dl_nseg * 3 * dl_ld));
  PGOC(dalloc(&dl_E2, (int64_t)dl_nseg * 3 * dl_ld));
  HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_chol_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::CHOL_LDS_BYTES));
  PGOC(dalloc(&dl_cvec, dl_Kp));
  PGOC(dalloc(&dl_ksep, 18 * std::max(1, dl_nsep)));
  PGOC(dalloc(&dl_R, std::max(1, dl_nU * dl_nU)));
  PGOC(dalloc(&dl_Wm, (int64_t)std::max(1, dl_nU) * dl_ld));
  PGOC(sync());  // the host lists die with this scope
  direct = true;
  dl_ready = true;
  return PGO_OK;
}

// (H + D'D) y = gs by Woodbury on chain + low rank, then iterative refinement against the assembled matrix; leaves the
// true residual in r (the model-decrease identity of lm_iteration_tail reads it) and |r|^2, |gs|^2 in scal[8..9].
int pgo_handle::direct_enqueue() {
  const int n = S.n_loc, K = dl_K, Kp = dl_Kp, nb = dl_Kp / 32;
  dev::DlrArgs A;
  A.n = n;
  A.m = dl_m;
  A.K = K;
  A.Kp = Kp;
  A.ld = dl_ld;
  A.jr = jr;
  A.scale = scale;
  A.d2 = d2;
  A.e_ia = e_ia;
  A.e_ib = e_ib;
  A.chain_edge = dl_chain_edge;
  A.lr_edge = dl_lr_edge;
  A.va = dl_va;
  A.vb = dl_vb;
  A.trec = dl_trec;
  A.fac = dl_fac;
  A.vrec = dl_vrec;
  A.Z = dl_Z;
  A.cap = dl_cap;
  A.dwork = dl_dwork;
  A.cvec = dl_cvec;
  A.nsep = dl_nsep;
  for (int j = 0; j < dev::DLR_MAX_SEP; ++j) A.sep[j] = dl_sep[j];
  A.ksep = dl_ksep;
  A.sw_js = has_sw ? sw_js : nullptr;
  A.sw_c = has_sw ? sw_c : nullptr;
  A.rec_n = rec_doubles;
  A.rec_info = info_mode ? 1 : 0;
  hipLaunchKernelGGL(dev::k_dlr_setup, dim3((n + dl_m + 255) / 256), dim3(256), 0, stream, A);
  PGOC(check_launch("k_dlr_setup"));
  hipLaunchKernelGGL(dev::k_dlr_factor, dim3(dl_nsep + 1), dim3(64), 0, stream, (const double*)dl_trec, n, dl_fac, A);
  PGOC(check_launch("k_dlr_factor"));
  hipLaunchKernelGGL(dev::k_dlr_prefix, dim3(1), dim3(128), 0, stream, (const double*)dl_fac, n, dl_nseg, dl_seglen, dl_pre);
  PGOC(check_launch("k_dlr_prefix"));
  const bool one_launch = dl_refine > 0 && dl_pre2 != nullptr;   // the refinement's single column: k_dlr_solve1
  if (one_launch) {
    hipLaunchKernelGGL(dev::k_dlr_prefix, dim3(1), dim3(512), 0, stream, (const double*)dl_fac, n, dl_nseg2, dl_seglen2, dl_pre2);
    PGOC(check_launch("k_dlr_prefix (fine segments)"));
  }
  dev::DlrColsArgs C;
  C.fac = dl_fac;
  C.pre = dl_pre;
  C.n = n;
  C.ncols = K + 1 + dl_nU;
  C.K = K;
  C.vec_col = K;
  C.ld = dl_ld;
  C.ucol0 = K + 1;
  C.nsep = dl_nsep;
  for (int j = 0; j < dev::DLR_MAX_SEP; ++j) C.sep[j] = dl_sep[j];
  C.ksep = dl_ksep;
  C.nseg = dl_nseg;
  C.seglen = dl_seglen;
  C.vrec = dl_vrec;
  C.va = dl_va;
  C.vb = dl_vb;
  C.rhs_b = gs;
  C.rhs_sub = nullptr;
  C.X = dl_Z;
  C.E = dl_E;
  C.E2 = dl_E2;
  auto solve_columns = [&](const dev::DlrColsArgs& Q) -> int {
    const dim3 grid((Q.ncols + 255) / 256, Q.nseg);
    hipLaunchKernelGGL(dev::k_dlr_fwd, grid, dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_mid, grid, dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_fix, grid, dim3(256), 0, stream, Q);
    return check_launch("k_dlr_fwd / _mid / _fix");
  };
  PGOC(solve_columns(C));
  // the couplings at the separators (k_dlr_sep_*): Y = the U columns of this solve, R once per factorisation
  dev::DlrSepArgs SA;
  SA.nsep = dl_nsep;
  SA.nU = dl_nU;
  SA.n = n;
  for (int j = 0; j < dev::DLR_MAX_SEP; ++j) SA.sep[j] = dl_sep[j];
  SA.ksep = dl_ksep;
  SA.Y = dl_Z + (K + 1);
  SA.yld = dl_ld;
  SA.Sinv = dl_R;
  SA.trec = dl_trec;
  SA.Wm = dl_Wm;
  auto separator_fix = [&](double* X, int ld, int ncols) -> int {
    if (dl_nsep == 0) return PGO_OK;
    dev::DlrSepArgs Q = SA;
    Q.X = X;
    Q.ld = ld;
    Q.ncols = ncols;
    hipLaunchKernelGGL(dev::k_dlr_sep_w, dim3((ncols + 255) / 256), dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_sep_apply, dim3((ncols + 255) / 256, (3 * n + 63) / 64), dim3(256), 0, stream, Q);
    return check_launch("k_dlr_sep_w / _apply");
  };
  if (dl_nsep > 0) {
    SA.X = dl_Z;
    SA.ld = dl_ld;
    SA.ncols = K + 1;
    hipLaunchKernelGGL(dev::k_dlr_sep_system, dim3(1), dim3(256), 0, stream, SA);
    PGOC(check_launch("k_dlr_sep_system"));
  }
  PGOC(separator_fix(dl_Z, dl_ld, K + 1));
  hipLaunchKernelGGL(dev::k_dlr_cap, dim3((std::max(Kp, K + 1) + 255) / 256, Kp), dim3(256), 0, stream, A);
  PGOC(check_launch("k_dlr_cap"));
  for (int kb = 0; kb < nb; ++kb) {
    hipLaunchKernelGGL(dev::k_chol_panel, dim3(std::max(1, nb - 1)), dim3(dev::CHOL_THREADS), dev::CHOL_LDS_BYTES, stream, dl_cap, dl_nm, dl_dwork, Kp, nb, kb);
    PGOC(check_launch("k_chol_panel"));
  }
  auto capacitance_solve = [&]() -> int {  // cvec <- (L L')^-1 cvec = N' (N cvec)
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)dl_nm, Kp, nb, (const double*)dl_cvec, dl_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply, dim3(nb), dim3(256), 0, stream, (const double*)dl_nm, Kp, nb, (const double*)dl_cy, dl_cvec, 1);
    return check_launch("k_tri_apply");
  };
  PGOC(capacitance_solve());
  hipLaunchKernelGGL(dev::k_dlr_combine, dim3((3 * n + 3) / 4), dim3(256), 0, stream, (const double*)dl_Z, dl_ld, K, (const double*)dl_cvec,
                     (const double*)dl_Z, dl_ld, K, 3 * n, y, 0);
  PGOC(check_launch("k_dlr_combine"));
  auto residual_product = [&]() -> int {  // ap = (H + D'D) y
    hipLaunchKernelGGL(dev::k_scatter_owned, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, y, p_full);
    PGOC(check_launch("k_scatter_owned"));
    return spmv_enqueue(p_full, ap, part[0], 1, nullptr);
  };
  for (int it = 0; it < dl_refine; ++it) {
    PGOC(residual_product());
    int xld = 64;   // layout of the single column in dl_x1: [3n][64] (batched kernels) or a plain vector (k_dlr_solve1)
    if (one_launch) {
      dev::DlrSolve1Args Q;
      Q.fac = dl_fac;
      Q.pre2 = dl_pre2;
      Q.n = n;
      Q.nseg = dl_nseg2;
      Q.seglen = dl_seglen2;
      Q.rhs_b = gs;
      Q.rhs_sub = ap;
      Q.x = dl_x1;
      Q.nsep = dl_nsep;
      Q.nU = dl_nU;
      for (int j = 0; j < dev::DLR_MAX_SEP; ++j) Q.sep[j] = dl_sep[j];
      Q.ksep = dl_ksep;
      Q.trec = dl_trec;
      Q.Y = dl_Z + (K + 1);
      Q.yld = dl_ld;
      Q.Sinv = dl_R;
      hipLaunchKernelGGL(dev::k_dlr_solve1, dim3(1), dim3(256), 0, stream, Q);
      PGOC(check_launch("k_dlr_solve1"));
      xld = 1;
    } else {
      dev::DlrColsArgs C1 = C;
      C1.ncols = 1;
      C1.K = 0;
      C1.vec_col = 0;
      C1.ld = 64;
      C1.rhs_sub = ap;
      C1.X = dl_x1;
      C1.nsep = 0;   // (no U columns: Y and R of the main solve are reused)
      PGOC(solve_columns(C1));
      PGOC(separator_fix(dl_x1, 64, 1));
    }
    hipLaunchKernelGGL(dev::k_dlr_vdot, dim3((Kp + 255) / 256), dim3(256), 0, stream, A, (const double*)dl_x1, xld, 0, dl_cvec);
    PGOC(check_launch("k_dlr_vdot"));
    PGOC(capacitance_solve());
    hipLaunchKernelGGL(dev::k_dlr_combine, dim3((3 * n + 3) / 4), dim3(256), 0, stream, (const double*)dl_Z, dl_ld, K, (const double*)dl_cvec,
                       (const double*)dl_x1, xld, 0, 3 * n, y, 1);
    PGOC(check_launch("k_dlr_combine"));
  }
  PGOC(residual_product());
  hipLaunchKernelGGL(dev::k_dlr_resid, dim3((3 * n + 255) / 256), dim3(256), 0, stream, (int64_t)3 * n, (const double*)gs, (const double*)ap, r);
  PGOC(check_launch("k_dlr_resid"));
  hipLaunchKernelGGL(dev::k_dot, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * n, (const double*)r, (const double*)r, part[2]);
  hipLaunchKernelGGL(dev::k_dot, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * n, (const double*)gs, (const double*)gs, part[4]);
  PGOC(check_launch("k_dot"));
  return reduce_to_scal({{part[2], g_flat, 0}, {part[4], g_flat, 0}}, 8);
}

// The ~130 launches of a direct solve are the same every time (every argument is fixed for the handle's lifetime; the
// trust-region radius enters through d2 on the device), so they can be captured once into a hipGraph and replayed with
// one host call -- measured: no gain in GN it/s (the solve is bound by its dependent kernels, not by the host), while
// capture + instantiation cost ~5 ms, as much as five LM iterations of a fresh handle.  Off unless PGO_DIRECT_GRAPH=1.
int pgo_handle::direct_solve() {
  if (opt.use_graphs && dl_use_graph && !dl_graph_failed) {
    if (!dl_graph_exec) {
      hipGraph_t gr = nullptr;
      HIPC(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      const int st_cap = direct_enqueue();
      const hipError_t e_end = hipStreamEndCapture(stream, &gr);
      hipError_t e_inst = hipSuccess;
      if (st_cap == PGO_OK && e_end == hipSuccess) {
        e_inst = hipGraphInstantiate(&dl_graph_exec, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
      }
      if (st_cap != PGO_OK || e_end != hipSuccess || e_inst != hipSuccess) {   // eager launches from here on
        (void)hipGetLastError();
        dl_graph_exec = nullptr;
        dl_graph_failed = true;
      }
    }
    if (dl_graph_exec) HIPC(hipGraphLaunch(dl_graph_exec, stream));
  }
  if (!dl_graph_exec) PGOC(direct_enqueue());
  if (dl_fail_at > 0 && iter == dl_fail_at)   // test hook ("direct_fail_at"): a direct solve that returns NaNs
    HIPC(hipMemsetAsync(y, 0xFF, (size_t)3 * S.n_loc * sizeof(double), stream));
  return PGO_OK;
}

// block-Jacobi PCG on (H + D2) y = gs, y0 = 0.  Host checks the residual every
// pcg_check_every iterations; in between the kernels early-out on st->done.
int pgo_handle::pcg(int* iters, double* rel) {
  if (solo) {
    dev::SoloProb P;
    P.row0 = 0;
    P.nrows = S.n_loc;
    P.tile0 = 0;
    P.ntiles = S.n_tiles();
    P.active = 1;
    P.max_it = std::max(0, opt.pcg_max_iters);
    P.rtol = opt.pcg_rtol;
    HIPC(hipMemcpyAsync(solo_prob, &P, sizeof P, hipMemcpyHostToDevice, stream));
    dev::SoloArgs A;
    A.A = spmv_args(p_full, ap, part[0], 1, nullptr);
    A.V = cg_vec();
    A.C = chain_pre();
    if (!chain_len) A.C.cw = nullptr;
    A.chain_steps = solo_steps;
    A.scan_levels = solo_scan;
    A.b = gs;
    A.prob = solo_prob;
    A.out = solo_out;
    A.x = poses;
    A.scale = scale;
    A.cand = cand;
    hipLaunchKernelGGL(dev::k_pcg_solo, dim3(1), dim3(dev::SOLO_WG), 0, stream, A);
    PGOC(check_launch("k_pcg_solo"));
    HIPC(hipMemcpyAsync(h_solo, solo_out, sizeof(dev::SoloOut), hipMemcpyDeviceToHost, stream));
    PGOC(sync());  // P (stack) was consumed by the copy above
    *iters = h_solo->iters;
    *rel = (h_solo->bb > 0.0) ? std::sqrt(h_solo->rr / h_solo->bb) : 0.0;
    last_pcg_iters = h_solo->iters;
    return PGO_OK;
  }
  dev::CgVec V = cg_vec();
  const bool multi = multi_rank();
  dev::GroupPre GP;
  GP.ginv = ginv;
  GP.B = grp_B;
  GP.nb = grp_nb;
  GP.nb_pad = grp_pad;
  GP.n_groups = n_groups;
  const bool grouped = grp_B > 1, chained = chain_len > 0;
  const int g_u1 = chained ? g_chain : (grouped ? g_grp : g_vec);  // grid (= number of partials) of the init / update1 kernels
  if (chained) launch_cg_init_chain(gs, part[0], part[1]);
  else if (grouped) hipLaunchKernelGGL(dev::k_cg_init_g, dim3(g_u1), dim3(dev::WG), 0, stream, V, GP, (const double*)gs, part[0], part[1]);
  else hipLaunchKernelGGL(dev::k_cg_init, dim3(g_u1), dim3(dev::WG), 0, stream, V, gs, part[0], part[1]);
  PGOC(check_launch("k_cg_init"));
  const bool sr = use_sr && chained;
  // single-reduction loop: "make u visible to the peers, w = A u, reduce (gamma, rr, delta) together, new coefficients"
  auto sr_product_and_scalars = [&](int first) -> int {
    int n_sp = g_spmv;
    const int32_t* done = first ? nullptr : &st->done;
    if (overlap) PGOC(spmv_with_halo(p_full, ap, part[2], done, &n_sp));
    else {
      PGOC(share_gather_vector(p_full));
      PGOC(spmv_enqueue(p_full, ap, part[2], 1, done));
    }
    PGOC(reduce_to_scal({{part[0], g_u1, 0}, {part[1], g_u1, 0}, {part[2], n_sp, 0}}, 4));
    hipLaunchKernelGGL(dev::k_cg_sr_scal, dim3(1), dim3(1), 0, stream, st, (const double*)(scal + 4), opt.pcg_rtol, first);
    return check_launch("k_cg_sr_scal");
  };
  if (sr) {
    // the start-up kernel left u = M^-1 b in the gather vector (and in z, which becomes p: beta = 0 in the first update)
    HIPC(hipMemsetAsync(sr_s, 0, (size_t)3 * S.n_loc * sizeof(double), stream));
    PGOC(sr_product_and_scalars(1));
  } else {
    // two levels: z (= p) of the start-up kernel gets the coarse correction, r.z one more partial
    const int n_rz0 = use_coarse ? g_u1 + co_ndot : g_u1;
    if (use_coarse) {
      PGOC(coarse_solve(part[0] + g_u1, nullptr));
      hipLaunchKernelGGL(dev::k_coarse_prolong, dim3((unsigned)std::min<int64_t>((S.n_loc + 255) / 256, 512)), dim3(256), 0, stream, (int)S.n_loc,
                         co_agg, (const double*)co_pb, (const double*)co_ec, z, p_full + dev::PS * (int64_t)S.lo, (const int32_t*)co_ok);
      PGOC(check_launch("k_coarse_prolong"));
    }
    PGOC(reduce_to_scal({{part[0], n_rz0, 0}, {part[1], g_u1, 0}}, 4));
    hipLaunchKernelGGL(dev::k_cg_init_fin, dim3(1), dim3(1), 0, stream, st, scal + 4, opt.pcg_rtol);
    PGOC(check_launch("k_cg_init_fin"));
    if (!overlap) PGOC(share_gather_vector(p_full));
  }
  const int max_it = std::max(0, opt.pcg_max_iters);
  int every = std::max(1, opt.pcg_check_every);
  // one PCG iteration = 3 dependent launches; `par` is the r.z double-buffer parity baked into the arguments
  const bool fused = fused_p && !multi;
  double* pbuf[2] = {p_full, p_full2};
  if (fused) HIPC(hipMemsetAsync(p_full2, 0, (size_t)dev::PS * n_full * sizeof(double), stream));  // "p_old" of iteration 0
  auto enqueue_iteration = [&](int par) -> int {
    if (sr) {
      launch_cg_sr_chain(V, part[0], part[1]);
      PGOC(check_launch("k_cg_sr_cl"));
      return sr_product_and_scalars(0);
    }
    int n_sp = g_spmv;
    dev::CgVec Vi = V;
    if (fused) {
      // the previous iteration's direction update happens inside this SpMV: p_old = pbuf[par ^ 1] -> p_new = pbuf[par]
      dev::SpmvArgs A = spmv_args(pbuf[par ^ 1], ap, part[0], 1, &st->done);
      A.z = z;
      A.p_new = pbuf[par];
      A.part_rz = part[1];
      A.part_rr = part[2];
      A.n_rz = A.n_rr = g_u1;
      A.parity = par ^ 1;
      A.st = st;
      hipLaunchKernelGGL(dev::k_spmv_t<5>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
      PGOC(check_launch("k_spmv (fused direction update)"));
      Vi.p = pbuf[par];
      Vi.fused = 1;
    } else if (overlap) PGOC(spmv_with_halo(p_full, ap, part[0], &st->done, &n_sp));  // p reaches the peers inside
    else PGOC(spmv_enqueue(p_full, ap, part[0], 1, &st->done));
    const double* pap = multi ? scal + 6 : part[0];
    const int n_pap = multi ? 1 : n_sp;
    if (multi) PGOC(reduce_to_scal({{part[0], n_sp, 0}}, 6));
    if (chained) launch_cg_update1_chain(Vi, par, pap, n_pap, part[1], part[2]);
    else if (grouped) hipLaunchKernelGGL(dev::k_cg_update1_g, dim3(g_u1), dim3(dev::WG), 0, stream, Vi, GP, par, pap, n_pap, part[1], part[2]);
    else hipLaunchKernelGGL(dev::k_cg_update1, dim3(g_u1), dim3(dev::WG), 0, stream, Vi, par, pap, n_pap, part[1], part[2]);
    PGOC(check_launch("k_cg_update1"));
    if (fused) return PGO_OK;  // its r.z / r.r partials are booked by the next SpMV, or by k_cg_book at the end of the slice
    if (use_coarse) {   // second level (single rank): e_c, its share of r.z as more partials, prolongation inside the direction update
      PGOC(coarse_solve(part[1] + g_u1, &st->done));
      hipLaunchKernelGGL(dev::k_cg_update2c, dim3(g_vec), dim3(dev::WG), 0, stream, V, par, (const double*)part[1], g_u1 + co_ndot,
                         (const double*)part[2], g_u1, co_agg, (const double*)co_pb, (const double*)co_ec);
      return check_launch("k_cg_update2c");
    }
    if (multi) {
      PGOC(reduce_to_scal({{part[1], g_u1, 0}, {part[2], g_u1, 0}}, 7));
      hipLaunchKernelGGL(dev::k_cg_update2, dim3(g_flat), dim3(dev::WG), 0, stream, V, par, scal + 7, 1, scal + 8, 1);
      PGOC(check_launch("k_cg_update2"));
      if (!overlap) PGOC(share_gather_vector(p_full));
    } else {
      hipLaunchKernelGGL(dev::k_cg_update2, dim3(g_flat), dim3(dev::WG), 0, stream, V, par, part[1], g_u1, part[2], g_u1);
      PGOC(check_launch("k_cg_update2"));
    }
    return PGO_OK;
  };
  // Launch-bound regime (small graphs): a slice of `every` iterations is captured ONCE per handle into a
  // hipGraph (every argument is fixed for the handle's lifetime; the slice length is even so the parity
  // pattern repeats) and replayed with a single host call per slice.
  // Several ranks: the slice is captured WITH its RCCL calls (all-reduces of the dot products, halo exchange or
  // all-gather of the search direction) when they are pure stream work and everything runs on the one solver stream
  // -- otherwise every PCG iteration costs the host ~7 kernel launches + 3 collective calls, about the device time of
  // an iteration at 8 shards of the 1M-pose graph.  PGO_GRAPH_COLLECTIVES=0 keeps the eager loop; a capture that fails
  // falls back to it for the rest of the handle's life.
  // The point-to-point halo exchange (an ncclSend / ncclRecv group) is NOT captured by default: it has never run against
  // a real peer (no multi-GPU lease yet), and a group of p2p calls inside a graph is the less travelled road -- its
  // first execution should be the plain one.  PGO_GRAPH_COLLECTIVES=2 captures it too.
  bool use_graph = opt.use_graphs && !graph_failed &&
                   (!multi || (comm->capturable() && !overlap && graph_collectives > 0 && (!use_halo || graph_collectives > 1)));
  if (use_graph) {
    every += every & 1;
    if (!cg_graph_exec || cg_graph_len != every) {
      if (cg_graph_exec) (void)hipGraphExecDestroy(cg_graph_exec);
      cg_graph_exec = nullptr;
      hipGraph_t gr = nullptr;
      HIPC(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
      int st_cap = PGO_OK;
      for (int c = 0; c < every && st_cap == PGO_OK; ++c) st_cap = enqueue_iteration(c & 1);
      if (fused && st_cap == PGO_OK) {
        hipLaunchKernelGGL(dev::k_cg_book, dim3(1), dim3(dev::WG), 0, stream, st, (every - 1) & 1, (const double*)part[1], g_u1, (const double*)part[2], g_u1);
        st_cap = check_launch("k_cg_book");
      }
      hipError_t e_end = hipStreamEndCapture(stream, &gr);
      hipError_t e_inst = hipSuccess;
      if (st_cap == PGO_OK && e_end == hipSuccess) {
        e_inst = hipGraphInstantiate(&cg_graph_exec, gr, nullptr, nullptr, 0);
        (void)hipGraphDestroy(gr);
      }
      if (st_cap != PGO_OK || e_end != hipSuccess || e_inst != hipSuccess) {
        if (!multi) {  // single rank: a capture failure is a real error
          PGOC(st_cap);
          if (e_end != hipSuccess) return fail(PGO_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e_end));
          return fail(PGO_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e_inst));
        }
        // with collectives inside: run eagerly from now on (every rank takes the same decision: same library, same calls)
        (void)hipGetLastError();
        cg_graph_exec = nullptr;
        graph_failed = true;
        use_graph = false;
        if (opt.verbose) printf("pgo: hipGraph capture of the PCG slice with collectives failed; eager launches from here on\n");
      } else {
        cg_graph_len = every;
      }
    }
  }
  // Slices enqueued after convergence are not free: each of their launches early-outs on st->done but still costs
  // ~4.5 us of device time (13 us per no-op PCG iteration at 1M poses -- with 100-iteration slices that was 5 % of an LM
  // iteration).  So the slices are short and the FIRST host check comes after as many of them as the previous solve of
  // this handle makes likely (85 % of its iteration count; consecutive LM iterations need similar counts), the
  // following checks after every slice.
  int it = 0;
  int ahead = (last_pcg_iters > 0) ? std::max(1, (int)(0.85 * last_pcg_iters) / every) : 1;
  while (true) {
    const double te0 = wall_s();
    const int it_before = it;
    for (int sl = 0; sl < ahead && it < max_it; ++sl) {
      const int chunk = std::min(every, max_it - it);
      if (use_graph && chunk == every && (it & 1) == 0) {
        HIPC(hipGraphLaunch(cg_graph_exec, stream));
      } else {
        for (int c = 0; c < chunk; ++c) PGOC(enqueue_iteration((it + c) & 1));
        if (fused && chunk > 0) {
          hipLaunchKernelGGL(dev::k_cg_book, dim3(1), dim3(dev::WG), 0, stream, st, (it + chunk - 1) & 1, (const double*)part[1], g_u1, (const double*)part[2], g_u1);
          PGOC(check_launch("k_cg_book"));
        }
      }
      it += chunk;
    }
    ahead = 1;
    t_enqueue += wall_s() - te0;
    n_enqueued += it - it_before;
    HIPC(hipMemcpyAsync(h_st, st, sizeof(dev::CgState), hipMemcpyDeviceToHost, stream));
    PGOC(sync());
    if (h_st->done || it >= max_it) break;
  }
  last_pcg_iters = h_st->iters;
  *iters = h_st->iters;
  *rel = (h_st->bb > 0.0) ? std::sqrt(h_st->rr / h_st->bb) : 0.0;
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

// the PCG preconditioner for the current LM diagonal: dense pose-group inverses or the chain factorisation
int pgo_handle::prepare_preconditioner() {
  if (grp_B > 1) {
    dev::GroupPrepArgs GA;
    GA.inc_ptr = inc_ptr;
    GA.inc_col = inc_col;
    GA.hoff = hoff;
    GA.hd = hd;
    GA.d2 = d2;
    GA.ginv = ginv;
    GA.n_loc = S.n_loc;
    GA.lo = S.lo;
    GA.B = grp_B;
    GA.nb = grp_nb;
    GA.n_groups = n_groups;
    hipLaunchKernelGGL(dev::k_prepare_groups, dim3(grp_prep_grid), dim3(dev::WG), grp_lds, stream, GA);
    PGOC(check_launch("k_prepare_groups"));
  }
  if (chain_len) PGOC(factor_chain());
  if (use_coarse) PGOC(coarse_factor());
  return PGO_OK;
}

// block LDL' of the chain preconditioner's segments (the records are complete: C part from k_assemble, M part from k_prepare)
int pgo_handle::factor_chain() {
  const int n_seg = (S.n_loc + chain_len - 1) / chain_len;
  // One THREAD per segment runs the recurrence (64 .. 256 dependent steps), so the kernel lives on memory requests in
  // flight, not on lanes: with 64 segments per wavefront the 15.6k segments of the 1M-pose graph are 244 wavefronts --
  // one per compute unit, 32 KB in flight each, 0.25 of the HBM roofline.  16 segments per wavefront (4 wavefronts per
  // compute unit, each with its own queue of outstanding loads) quadruple that.
  int spw = 64;
  while (spw > 8 && n_seg / spw < 1024) spw >>= 1;
  if ((chain_chunk ? chain_chunk : dev::CHAIN_CHUNK) == 2)
    hipLaunchKernelGGL(dev::k_chain_factor<2>, dim3((n_seg + spw - 1) / spw), dim3(spw), 0, stream, (const double*)chain_c, S.n_loc, chain_pad,
                       chain_len, chain_w, chain_s);
  else
    hipLaunchKernelGGL(dev::k_chain_factor<4>, dim3((n_seg + spw - 1) / spw), dim3(spw), 0, stream, (const double*)chain_c, S.n_loc, chain_pad,
                       chain_len, chain_w, chain_s);
  return check_launch("k_chain_factor");
}

// LM diagonal for the current radius (per problem in a batched handle) + the preconditioner's set-up
int pgo_handle::prepare_system() {
  hipLaunchKernelGGL(dev::k_prepare, dim3(g_rows), dim3(dev::WG), 0, stream, hd, (const double*)diag_full, S.n_loc, S.lo, fixed_internal, radius,
                     opt.min_lm_diagonal, opt.max_lm_diagonal, d2, minv, (const uint8_t*)fixed_mask, (const int32_t*)prob_of_256,
                     (const double*)prob_radius, chain_len ? chain_c : (double*)nullptr, hdd);
  PGOC(check_launch("k_prepare"));
  if (!direct) PGOC(prepare_preconditioner());   // (the direct solve does not need it; its PCG fallback sets it up on demand)
  return PGO_OK;
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
      hipLaunchKernelGGL(dev::k_scatter_owned, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, y, p_full);
      PGOC(check_launch("k_scatter_owned"));
      PGOC(share_gather_vector(p_full));
    }
    if (true_residual) {
      PGOC(spmv_enqueue(p_full, ap, part[0], 1, nullptr));
      hipLaunchKernelGGL(dev::k_dlr_resid, dim3((3 * S.n_loc + 255) / 256), dim3(256), 0, stream, (int64_t)3 * S.n_loc, (const double*)gs,
                         (const double*)ap, r);
      PGOC(check_launch("k_dlr_resid"));
    }
    // y.(H y) = y.b - y.r - y.(D y) from the residual (no further product by H): part[1] = y.b, part[0] = y.r, part[5] = y.(D y)
    hipLaunchKernelGGL(dev::k_model_terms, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * S.n_loc, (const double*)y, (const double*)gs,
                       (const double*)r, (const double*)d2, part[1], part[0], part[5]);
    PGOC(check_launch("k_model_terms"));
    // candidate x + d and |d|^2
    double* x_old = poses;
    hipLaunchKernelGGL(dev::k_candidate, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, x_old, scale, y, cand, part[3]);
    PGOC(check_launch("k_candidate"));
  }
  double model_sw = 0.0, step2_sw = 0.0;
  if (has_sw) {  // back-substitute the switches (needs y of both endpoints: in the gather vector after the share above)
    const int g_sw = std::min(std::max(1, (S.n_edges_local + dev::WG - 1) / dev::WG), 1024);
    hipLaunchKernelGGL(dev::k_switch_backsub, dim3(g_sw), dim3(dev::WG), 0, stream, switch_arrays(), (const int32_t*)e_ia,
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
    hipLaunchKernelGGL(dev::k_xnorm, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, poses, scale, part[1]);
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
      hipLaunchKernelGGL(dev::k_grad_max, dim3(g_flat), dim3(dev::WG), 0, stream, gs, scale, S.n_loc, S.lo, part[0]);
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

// ====================================================================== C-ABI
static int require_device(int device) {
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
    hipLaunchKernelGGL(dev::k_edge_chi2, dim3(grid), dim3(dev::WG), 0, h->stream, A, (const int32_t*)h->e_orig, h->chi2_buf);
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
  out->n_incidences = h->S.n_inc;
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
    hipLaunchKernelGGL(dev::k_prob_reduce, dim3(n), dim3(dev::WG), 0, H.stream, (const dev::ProbRange*)d_range,
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
  hipLaunchKernelGGL(dev::k_jacobi_scale, dim3(H.g_rows), dim3(dev::WG), 0, H.stream, H.hd, H.S.n_loc, H.S.lo, -1, 0, H.scale,
                     (const uint8_t*)H.fixed_mask);
  PGOC(H.check_launch("k_jacobi_scale"));
  PGOC(H.eval_enqueue(H.poses, nullptr, 1, true, 0));
  PGOC(H.assemble_enqueue());
  if (o.jacobi_scaling) {
    hipLaunchKernelGGL(dev::k_jacobi_scale, dim3(H.g_rows), dim3(dev::WG), 0, H.stream, H.hd, H.S.n_loc, H.S.lo, -1, 1, H.scale,
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
  hipLaunchKernelGGL(dev::k_pcg_solo, dim3(n), dim3(dev::SOLO_WG), 0, H.stream, A);
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
    hipLaunchKernelGGL(dev::k_accept_rows, dim3(H.g_flat), dim3(dev::WG), 0, H.stream, H.S.n_loc, H.S.lo, (const int32_t*)H.prob_of_256,
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
  hipLaunchKernelGGL(dev::k_jacobi_scale, dim3(h->g_rows), dim3(dev::WG), 0, h->stream, h->hd, h->S.n_loc, h->S.lo,
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
  hipLaunchKernelGGL(dev::k_scatter_owned, dim3(h->g_flat), dim3(dev::WG), 0, h->stream, h->S.n_loc, h->S.lo, h->y, h->p_full);
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
    hipLaunchKernelGGL(dev::k_cg_init_g, dim3(h->g_grp), dim3(dev::WG), 0, h->stream, V, GP, (const double*)h->ap, h->part[0], h->part[1]);
  } else {
    hipLaunchKernelGGL(dev::k_cg_init, dim3(h->g_vec), dim3(dev::WG), 0, h->stream, V, (const double*)h->ap, h->part[0], h->part[1]);
  }
  PGOC(h->check_launch("k_cg_init (debug)"));
  if (h->use_coarse) {   // the second level's share of z
    PGOC(h->coarse_solve(h->part[3], nullptr));
    hipLaunchKernelGGL(dev::k_coarse_prolong, dim3((unsigned)std::min<int64_t>((h->S.n_loc + 255) / 256, 512)), dim3(256), 0, h->stream,
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
  out->algorithmic_bytes = 112.0 * h->S.n_edges_local + 80.0 * (double)h->S.n_inc + (100.0 + (h->chain_len ? 72.0 : 0.0)) * h->S.n_loc;
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
  out->units = h->S.n_inc + h->S.n_loc;
  // 76 B per off-diagonal block (value + column) ; per row: 48 B diagonal planes + 24 B D'D + 4 B row
  // pointer + 24 B y + 24 B p; the product kernel k_spmv_p reads the diagonal with D'D folded in (k_prepare): 24 B less
  out->algorithmic_bytes = 76.0 * (double)h->S.n_inc + (h->spmv_pipe ? 100.0 : 124.0) * h->S.n_loc;
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
      hipLaunchKernelGGL(dev::k_cg_init_g, dim3(h->g_grp), dim3(dev::WG), 0, h->stream, V, GP, (const double*)h->gs, h->part[0], h->part[1]);
    }, &ms));
    out->algorithmic_bytes = (8.0 * 3 * h->grp_nb + 24.0 + 4 * 24.0) * nl;
  } else {
    PGOC(time_launches(h, reps, [&] {
      hipLaunchKernelGGL(dev::k_cg_init, dim3(h->g_vec), dim3(dev::WG), 0, h->stream, V, (const double*)h->gs, h->part[0], h->part[1]);
    }, &ms));
    out->algorithmic_bytes = (48.0 + 24.0 + 4 * 24.0) * nl;
  }
  out->ms_avg = ms;
  out->units = h->S.n_loc;
  return PGO_OK;
}

}  // extern "C"
