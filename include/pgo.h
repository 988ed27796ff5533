/*
 * pgo.h -- C-ABI of the MI355X-native 2D pose-graph backend (libpgo.so).
 *
 * This is the drop-in boundary for ONE path of wei-ght/toy-robust-backend-slam:
 * DCS-ceres/main.cpp METHOD 0/1 (SE(2) odometry + loop-closure least squares,
 * Dynamic Covariance Scaling, HuberLoss(0.01), Ceres LM) and, since SURVEY.md section 8(f) ranks it
 * next, METHOD 2 (switchable constraints).  Everything the
 * reference does between `ceres::Problem problem;` (main.cpp:66) and the end of
 * `ceres::Solve` (main.cpp:163) is replaced by pgo_create / pgo_solve /
 * pgo_get_poses; the g2o loader, classifier, outlier injector and writers
 * (include/g2o_util.h:23-186) are replaced by the pgo_g2o_* / pgo_inject_* /
 * pgo_write_* host functions.  All citations are relative to /root/reference/DCS-ceres.
 *
 * Conventions
 *   - plain pointers + sizes only; the caller owns every host array it passes;
 *     the library copies in at create and copies out on get.
 *   - every function returns 0 on success or a negative pgo_status.
 *     No exception crosses this boundary.
 *   - a pgo_t handle is NOT thread-safe: one handle per host thread.
 *   - functions marked [host] never touch the GPU and work on a GPU-less box;
 *     functions marked [gpu] require a gfx950 device and fail with
 *     PGO_ERR_NO_DEVICE otherwise (there is no CPU fallback on the product path).
 */
#ifndef PGO_H_
#define PGO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum pgo_status {
  PGO_OK = 0,
  PGO_ERR_INVALID_ARG = -1,   /* null pointer, bad size, bad index            */
  PGO_ERR_IO = -2,            /* file cannot be opened / written              */
  PGO_ERR_PARSE = -3,         /* malformed g2o record                         */
  PGO_ERR_NO_DEVICE = -4,     /* no gfx950 device visible                     */
  PGO_ERR_HIP = -5,           /* a HIP runtime call failed                    */
  PGO_ERR_COMM = -6,          /* RCCL / shm communicator failure              */
  PGO_ERR_NUMERIC = -7,       /* non-finite residual/Jacobian at the current point
                                 (Ceres: "Residual and Jacobian evaluation failed") */
  PGO_ERR_UNSUPPORTED = -8,   /* METHOD 3/4 etc.                              */
  PGO_ERR_NOMEM = -9
} pgo_status;

const char* pgo_strerror(int status);           /* [host] static string           */
const char* pgo_last_error(void);               /* [host] thread-local detail text */
const char* pgo_version(void);                  /* [host]                          */

/* edge kinds: include/g2o_util.h:14-16 */
#define PGO_EDGE_ODOMETRY 0
#define PGO_EDGE_CLOSURE 1
#define PGO_EDGE_BOGUS 2

/* ------------------------------------------------------ g2o graph (host side)
 * Replaces class ReadG2O (include/g2o_util.h:20-188) + Node/Edge (include/graph.h).
 * The graph is held as flat arrays; edges are stored in the reference's
 * residual-block order: all odometry, then all closure, then all bogus
 * (main.cpp:95-150).                                                          */
typedef struct pgo_graph pgo_graph;

/* ReadG2O::ReadG2O (g2o_util.h:23-89).  Same tags (VERTEX_SE2|VERTEX2,
 * EDGE_SE2|EDGE2), same positional fields, same classifier: odometry iff
 * abs(a-b) < 5 (g2o_util.h:68), endpoints addressed by vector position
 * (g2o_util.h:70,77).  Unlike the reference, I/O and range errors are reported. */
int pgo_g2o_load(const char* path, pgo_graph** out);                       /* [host] */
/* same, from a memory buffer (used by tests and by the synthetic generator)      */
int pgo_g2o_parse(const char* text, size_t len, pgo_graph** out);          /* [host] */
/* build a graph directly from arrays (kind[] decides the three lists)            */
int pgo_graph_from_arrays(int32_t n_poses, const double* poses_xyt,
                          int32_t n_edges, const int32_t* ia, const int32_t* ib,
                          const double* meas_xyt, const double* info6_or_null,
                          const uint8_t* kind, pgo_graph** out);           /* [host] */
void pgo_graph_free(pgo_graph* g);                                         /* [host] */

int32_t pgo_graph_num_poses(const pgo_graph* g);
int32_t pgo_graph_num_edges(const pgo_graph* g);            /* odo + closure + bogus */
int32_t pgo_graph_num_edges_of_kind(const pgo_graph* g, int kind);
/* borrowed pointers, valid until the graph is mutated or freed                    */
const int32_t* pgo_graph_pose_ids(const pgo_graph* g);      /* Node::index          */
double*        pgo_graph_poses(pgo_graph* g);               /* N x 3 (x,y,theta), mutable: Node::p */
const int32_t* pgo_graph_edge_a(const pgo_graph* g);        /* position of Edge::a  */
const int32_t* pgo_graph_edge_b(const pgo_graph* g);
const double*  pgo_graph_edge_meas(const pgo_graph* g);     /* E x 3 (x,y,theta)    */
const double*  pgo_graph_edge_info(const pgo_graph* g);     /* E x 6 I11 I12 I13 I22 I23 I33 (parsed, unused: SURVEY H5) */
const uint8_t* pgo_graph_edge_kind(const pgo_graph* g);

/* ReadG2O::add_random_C (g2o_util.h:151-171): `count` bogus loops drawn with the
 * C library rand() in the reference's call order (a, b, m0, m1, m2), measurement
 * rand()/RAND_MAX in INTEGER arithmetic, info (2,0,0,300,0,300).
 * seed >= 0: srand(seed) first (reproducible);  seed < 0: srand(time(0)) as
 * main.cpp:43 does.                                                              */
int pgo_inject_outliers(pgo_graph* g, int32_t count, int64_t seed);        /* [host] */

/* writePoseGraph_nodes / writePoseGraph_edges (g2o_util.h:93-112,179-186).
 * precision <= 0 : the reference's default ostream formatting (6 significant
 * digits); precision > 0 : that many significant digits (17 round-trips).        */
int pgo_write_nodes(const pgo_graph* g, const char* path, int precision);  /* [host] */
int pgo_write_edges(const pgo_graph* g, const char* path);                 /* [host] */
/* writePoseGraph_switches (g2o_util.h:114-148): three sections, "<a> <b> <type> <prior> <switch>" per edge;
 * switches: E values in the graph's edge order (prior is 1.0 everywhere, as main.cpp:119,141)             */
int pgo_write_switches(const pgo_graph* g, const char* path, const double* switches); /* [host] */
/* g2o writer (VERTEX_SE2 / EDGE_SE2), for the synthetic configs                  */
int pgo_write_g2o(const pgo_graph* g, const char* path);                   /* [host] */

/* Synthetic Manhattan world (BASELINE configs C4/C5; the reference ships no
 * generator -- spec in SURVEY.md section 8(d)): unit steps on the integer grid of a
 * bounded square (side ~ sqrt(N/4), so cells are revisited), +-90 degree turns with
 * p=0.2, odometry noise N(0, 0.02 m / 0.01 rad), up to 3 closures per pose to earlier
 * poses (|i-j| >= 5) within 1.5 m, thinned to about edges_per_pose * N edges in total,
 * initial poses = dead-reckoned odometry, plus round(outlier_frac * #closures) bogus
 * loops with R4 semantics (uniform random endpoints, zero measurement).
 * PRNG: splitmix64(seed).                                                         */
int pgo_synth_manhattan(int32_t n_poses, double edges_per_pose, double outlier_frac,
                        uint64_t seed, pgo_graph** out);                   /* [host] */

/* ------------------------------------------------------------ solver options
 * Defaults (pgo_options_default) are the Ceres defaults the reference runs with
 * (main.cpp:154-163 sets only progress + SPARSE_NORMAL_CHOLESKY) plus the
 * constants hard-coded in the reference: Huber 0.01 (main.cpp:68), phi 0.5
 * (src/ceres_error.cpp:185), fixed pose 0 (main.cpp:153).                        */
typedef struct pgo_options {
  int32_t method;              /* 0 = plain (OdometryResidue everywhere), 1 = DCS on closure+bogus (main.cpp:112-114,135-137),
                                  2 = switchable constraints on closure+bogus (main.cpp:115-125,138-145) */
  int32_t max_iters;           /* 50   Solver::Options::max_num_iterations        */
  int32_t fixed_pose;          /* 0    position of the constant pose, -1 = none   */
  int32_t jacobi_scaling;      /* 1                                               */
  double  phi;                 /* 0.5  DCS upper bound                            */
  double  huber_delta;         /* 0.01 ; <= 0 disables the loss                   */
  double  ftol;                /* 1e-6  function_tolerance                        */
  double  gtol;                /* 1e-10 gradient_tolerance (max-norm)             */
  double  ptol;                /* 1e-8  parameter_tolerance                       */
  double  radius0;             /* 1e4   initial_trust_region_radius               */
  double  max_radius;          /* 1e16                                            */
  double  min_radius;          /* 1e-32                                           */
  double  min_relative_decrease; /* 1e-3                                          */
  double  min_lm_diagonal;     /* 1e-6                                            */
  double  max_lm_diagonal;     /* 1e32                                            */
  /* linear solver: block-Jacobi preconditioned CG on (J'J + D'D) y = J'r        */
  double  pcg_rtol;            /* stop when ||r|| <= pcg_rtol * ||b||  (1e-12: "exact" mode
                                  standing in for SPARSE_NORMAL_CHOLESKY; 0.1 = Ceres' eta for inexact steps) */
  int32_t pcg_max_iters;       /* cap per LM iteration                            */
  int32_t pcg_check_every;     /* iterations enqueued between host residual checks */
  int32_t verbose;             /* 1 = Ceres-like per-iteration table on stdout    */
  int32_t use_graphs;          /* 1 (default) = replay slices of pcg_check_every PCG iterations as a hipGraph (world == 1) */
  int32_t pcg_block_poses;     /* poses per block of the block-Jacobi preconditioner: 1 = the 3x3 pose blocks,
                                  2..32 = dense (3B x 3B) blocks of B consecutive poses (explicit inverses);
                                  0 = auto (32 for graphs of <= 8192 poses, which are launch-latency bound and
                                  chain-like, else 4) */
  int32_t halo_exchange;       /* world > 1: how the search direction reaches the other ranks each PCG iteration.
                                  0 (default) = in-place all-gather of all 3N doubles (exercised through RCCL, captured into
                                      the PCG hipGraph);
                                  1 (opt-in) = point-to-point halo exchange: every rank sends each peer only the rows that
                                      peer's off-diagonal blocks reference (one ncclSend/ncclRecv group on the solver's stream).
                                      Far fewer bytes, but the RCCL send/recv group has not yet run against a real peer:
                                      bench.py checks it against the all-gather result when it runs on several GPUs and
                                      times it only if the two agree                                                  */
  double  sc_prior_lambda;     /* 1.0  METHOD 2: weight of the switch prior sqrt(lambda)(1 - s)  (main.cpp:107)    */
  int32_t pose_ordering;       /* internal numbering of the poses (results are always in the caller's numbering):
                                  0 = the caller's, 1 = locality ordering (pgo_pose_order: segments of 64 consecutive
                                  poses reordered by reverse Cuthill-McKee on the loop edges that a neighbouring edge
                                  supports), -1 (default) = 1 when world > 1 (it shrinks every rank's halo) and on single-rank
                                  graphs of more than 65536 poses (the gathers of K1 / K2 / K3 then hit the XCD's L2), else 0 */
  int32_t info_weighting;      /* 0 (default) = the reference's objective: the information entries of an edge are parsed
                                  but unused.  1 = optional mode: every residual is whitened by
                                  its information matrix, e_w = L' e with Omega = L L', so |e_w|^2 = e' Omega e (the chi2
                                  of compute_edge_mahalanobis, src/layer_manager.cpp:230-282); Huber then acts on chi2 and
                                  METHOD 1 uses the chi2 form of DCS, s = min(1, 2 phi / (phi + chi2)), e = s e_w
                                  (differentiated through s, as the reference differentiates through psi).  Needs the
                                  information matrices (pgo_create_weighted / pgo_create_from_graph), all positive
                                  definite; METHOD 2: PGO_ERR_UNSUPPORTED                                            */
  int32_t pcg_chain_len;       /* chain preconditioner: block-Jacobi over segments of this many consecutive poses whose blocks
                                  are kept block-TRIDIAGONAL (the odometry chain inside the segment; every other edge only
                                  adds its 3x3 diagonal blocks), factorised exactly per LM iteration and applied by
                                  chunked wavefront scans.  A multiple of 4 that divides 256 (64 = the measured default)
                                  turns it on and overrides pcg_block_poses; 0 = off;
                                  -1 (default), with pcg_block_poses = 0 (auto): 64 on graphs of > 50000 poses; 256 on chain-like
                                  graphs of 512..8192 poses (few short-range non-consecutive edges); else off                 */
  int32_t halo_overlap;        /* 0 (default): exchange, then one SpMV launch, all on the solver's stream.
                                  1 (opt-in): with halo_exchange = 1, the exchange runs on a second stream while the SpMV
                                  multiplies the blocks whose columns are owned; the blocks that need halo rows follow.
                                  Checked against the plain schedule with the host-staged test communicator only: the
                                  RCCL send/recv group has not yet run against a real peer                              */
  int32_t linear_solver;       /* how (J'J + D'D) y = J'r is solved (the reference: SPARSE_NORMAL_CHOLESKY, main.cpp:154-163):
                                  1 = block-Jacobi PCG to pcg_rtol;
                                  2 = direct: the odometry chain (one edge per consecutive pose pair; block tridiagonal, factorised
                                      exactly) + every other edge as a low-rank term through the Woodbury identity -- a dense
                                      Cholesky of order 3 x (edges outside the chain) -- + iterative refinement.  One rank, no
                                      information weighting, a constant pose, every consecutive pose pair joined by an edge, at most 2047
                                      edges outside the chain, at most 65536 poses; else PGO_ERR_UNSUPPORTED;
                                  0 (default) = auto, when pcg_rtol <= 1e-8 (the "exact" mode) and pcg_block_poses / pcg_chain_len
                                      are left at auto: 2 where it applies with at most 682 edges outside the chain (INTEL,
                                      MIT, CSAIL, FR079 ...); with more (M3500, FRH: the dense Cholesky is no longer cheap) the
                                      solve starts with 1 and changes to 2 after an LM iteration whose PCG iteration count
                                      says PCG costs more than the direct solve (M3500 with DCS: yes, without: no); else 1      */
  int32_t pcg_coarse_poses;    /* second preconditioner level: an additive coarse correction on the RIGID-BODY modes (translation x, y,
                                  rotation about the centre) of aggregates of this many consecutive poses -- three unknowns per
                                  aggregate, Galerkin matrix P'(J'J + D'D)P factorised densely per LM iteration (order <= 6143).
                                  It removes the smooth long-range error that block-Jacobi cannot: M3500 METHOD 1, PCG to 1e-10:
                                  1557 -> 178 iterations with 16-pose aggregates.  Rounded up to a multiple of the one-level
                                  block (pcg_chain_len / pcg_block_poses).  One rank.  0 = off;
                                  -1 (default) = auto, for graphs of >= 512 poses that stay on PCG while pcg_block_poses and
                                  pcg_chain_len are left at auto: on for tight solves
                                  (pcg_rtol <= 1e-3) -- 16 poses per aggregate up to 8192 poses, else 64, doubled until the
                                  coarse order fits; loose solves (the inexact mode) stay on one level unless asked        */
} pgo_options;

void pgo_options_default(pgo_options* o);                                  /* [host] */

typedef enum pgo_termination {
  PGO_TERM_CONVERGENCE_FTOL = 1,
  PGO_TERM_CONVERGENCE_GTOL = 2,
  PGO_TERM_CONVERGENCE_PTOL = 3,
  PGO_TERM_NO_CONVERGENCE = 4,       /* max_iters reached                        */
  PGO_TERM_MIN_RADIUS = 5,
  PGO_TERM_FAILURE = 6
} pgo_termination;

typedef struct pgo_iter_record {     /* one row of Ceres' progress table          */
  int32_t iter;
  int32_t step_ok;                   /* 1 accepted, 0 rejected, -1 invalid        */
  double  cost;
  double  cost_change;
  double  gradient_max_norm;
  double  step_norm;
  double  relative_decrease;         /* tr_ratio                                  */
  double  radius;
  int32_t pcg_iters;
  int32_t _pad;
  double  pcg_rel_residual;
  double  seconds;
} pgo_iter_record;

typedef struct pgo_summary {
  int32_t termination;               /* pgo_termination                           */
  int32_t iterations;                /* LM iterations performed (successful + not) */
  int32_t successful_steps;
  int32_t total_pcg_iters;
  double  initial_cost;
  double  final_cost;
  double  seconds_total;
  double  seconds_eval;              /* residual + Jacobian kernel                */
  double  seconds_assemble;
  double  seconds_linear;
  double  seconds_candidate;
} pgo_summary;

/* ------------------------------------------------------------ communicator
 * One process per GPU.  The graph is sharded by pose-id range over the ranks of
 * a communicator; world == 1 needs no communicator (pass NULL to pgo_create).   */
typedef struct pgo_comm pgo_comm;
#define PGO_COMM_ID_BYTES 128
int  pgo_comm_unique_id(uint8_t id[PGO_COMM_ID_BYTES]);        /* [gpu] ncclGetUniqueId on rank 0; broadcast by the caller */
int  pgo_comm_create_rccl(const uint8_t id[PGO_COMM_ID_BYTES], int rank, int world, int device, pgo_comm** out); /* [gpu] */
/* host-staged shared-memory communicator: TEST backend only (several ranks on one
 * GPU, where RCCL refuses duplicate devices).  Same collectives, same results.   */
int  pgo_comm_create_shm(const char* name, int rank, int world, int device, pgo_comm** out);                     /* [gpu] */
void pgo_comm_destroy(pgo_comm* c);

/* ------------------------------------------------------------------ solver
 * pgo_create replaces main.cpp:66-68,95-153: it takes the whole graph (every
 * rank passes the same arrays) and keeps the shard of `comm`'s rank on `device`. */
typedef struct pgo_handle pgo_t;

int pgo_create(pgo_t** h, int32_t n_poses, const double* poses_xyt,
               int32_t n_edges, const int32_t* ia, const int32_t* ib,
               const double* meas_xyt, const uint8_t* kind,
               const pgo_options* opt, pgo_comm* comm_or_null, int device);       /* [gpu] */
/* same, with the edges' information matrices: info6 = E x 6 (I11 I12 I13 I22 I23 I33, the reference's Edge fields,
 * include/graph.h:41-47) or NULL.  They are used by opt->info_weighting = 1 and by pgo_edge_chi2 only.                */
int pgo_create_weighted(pgo_t** h, int32_t n_poses, const double* poses_xyt,
                        int32_t n_edges, const int32_t* ia, const int32_t* ib,
                        const double* meas_xyt, const double* info6_or_null, const uint8_t* kind,
                        const pgo_options* opt, pgo_comm* comm_or_null, int device); /* [gpu] */
/* passes the graph's information matrices along */
int pgo_create_from_graph(pgo_t** h, const pgo_graph* g, const pgo_options* opt,
                          pgo_comm* comm_or_null, int device);                    /* [gpu] */
void pgo_destroy(pgo_t* h);

/* Problem::Evaluate equivalent.  poses_or_null == NULL evaluates at the handle's
 * current poses.  r: E x 3, J: E x 18 = [d e/d P1 (3x3 row-major) | d e/d P2],
 * both in the caller's edge order.  apply_loss != 0 applies the Huber corrector
 * (r <- sqrt(rho') r, J <- sqrt(rho') J) as Ceres' ResidualBlock::Evaluate does.
 * cost = 1/2 sum rho(|e|^2) (always with the loss when huber_delta > 0).
 * r/J outputs need world == 1.                                                  */
int pgo_eval(pgo_t* h, const double* poses_or_null, int apply_loss,
             double* cost, double* r_or_null, double* J_or_null);                 /* [gpu] */

/* compute_edge_mahalanobis (src/layer_manager.cpp:230-282; the layer managers' edge gate) for every edge at once:
 * chi2[e] = r' Omega r of the PLAIN residual r = (ex, ey, asin(clamp(sin delta, -1, 1))), clamped at 0, in the
 * caller's edge order, at the handle's current poses or at poses_or_null.  Independent of opt->method and
 * opt->info_weighting; any symmetric Omega.  Needs a handle created with information matrices.
 * world > 1: every rank gets the whole vector (one all-reduce).                                                */
int pgo_edge_chi2(pgo_t* h, const double* poses_or_null, double* chi2_out /* E */);   /* [gpu] */

/* ceres::Solve (main.cpp:163): LM from the current poses for opt.max_iters.      */
int pgo_solve(pgo_t* h, pgo_summary* s);                                          /* [gpu] */
/* Many independent problems at once (the reference's layer managers run ceres::Solve per candidate layer / window,
 * src/simple_layer_manager.cpp:457-622, src/layer_manager.cpp:104-179): pgo_solve on each of the n handles, driven by
 * up to max_concurrency host threads (<= 0: 8).  Every handle has its own HIP stream, so the launch-bound small solves
 * overlap on the device; each handle's result is bitwise what pgo_solve alone gives.  summaries: n entries or NULL.
 * Handles with a communicator are refused (PGO_ERR_UNSUPPORTED).  Returns the first failing status.                */
int pgo_solve_batch(pgo_t* const* handles, int32_t n, pgo_summary* summaries, int32_t max_concurrency);   /* [gpu] */
/* The batch as ONE handle (what the layer managers' evaluate_layer_cost / optimize_layer / optimize_local_window loops
 * want, src/simple_layer_manager.cpp:457-622): the block-diagonal union of n independent problems.  One launch of the
 * fused edge kernel, of the assembly kernel and of the preconditioner set-up covers every problem; each problem's linear
 * system is solved by its own workgroup in a single launch (the whole PCG solve, device-resident scalars); radius, cost,
 * accept / reject and termination are kept per problem and every problem stops by its own tests.  About ten launches per
 * LM iteration for the whole batch.  Every problem's result equals what pgo_solve gives for it alone up to the
 * association of floating-point sums.  One set of options for all problems: METHOD 0 or 1, opt->fixed_pose = the
 * constant pose of EVERY problem (its own numbering), preconditioner = chain segments (what the library would choose for
 * the largest problem alone; 64-pose segments where that would be dense pose blocks) or pcg_block_poses = 1.
 * Not supported (PGO_ERR_UNSUPPORTED): METHOD 2, info_weighting, a row with more than 256 incident edges, a communicator. */
typedef struct pgo_batch pgo_batch_t;
int pgo_batch_create(pgo_batch_t** b, int32_t n_problems, const pgo_graph* const* graphs,
                     const pgo_options* opt, int device);                          /* [gpu] */
void pgo_batch_destroy(pgo_batch_t* b);
int32_t pgo_batch_size(const pgo_batch_t* b);
/* ceres::Solve on every problem; summaries: n entries or NULL */
int pgo_batch_solve(pgo_batch_t* b, pgo_summary* summaries);                      /* [gpu] */
int pgo_batch_get_poses(pgo_batch_t* b, int32_t problem, double* out_xyt);        /* [gpu] */
int pgo_batch_set_poses(pgo_batch_t* b, int32_t problem, const double* poses_xyt); /* [gpu] */
int32_t pgo_batch_num_iter_records(const pgo_batch_t* b, int32_t problem);
int pgo_batch_get_iter_records(const pgo_batch_t* b, int32_t problem, pgo_iter_record* out, int32_t cap);

/* the same minimiser, resumable: (re)start with pgo_lm_begin, then run LM
 * iterations in slices (bench.py times slices); returns *done != 0 once a
 * termination test fired.                                                        */
int pgo_lm_begin(pgo_t* h);                                                       /* [gpu] */
int pgo_lm_step(pgo_t* h, int32_t n_iters, int32_t* done, pgo_summary* s);        /* [gpu] */
int32_t pgo_num_iter_records(const pgo_t* h);
int pgo_get_iter_records(const pgo_t* h, pgo_iter_record* out, int32_t cap);

/* What the handle resolved its "auto" options to and the size of its shard (bench.py / reports read this instead of
 * repeating the library's rules).                                                                                   */
typedef struct pgo_handle_info {
  int32_t n_poses, n_edges;          /* the whole graph                                                             */
  int32_t world, rank;
  int32_t row_lo, row_hi;            /* owned rows, internal numbering                                              */
  int32_t n_edges_local;             /* edges touching an owned row                                                 */
  int32_t n_tiles;                   /* row tiles of K2 / K3                                                        */
  int64_t n_incidences;              /* off-diagonal blocks of the owned rows                                       */
  int32_t pcg_block_poses;           /* resolved: poses per dense preconditioner block (1 when the chain form is on) */
  int32_t pcg_chain_len;             /* resolved: segment length of the chain preconditioner, 0 = off               */
  int32_t chain_kernel;              /* 0 = scan form, 2 / 4 = lean form with that many poses per lane              */
  int32_t pose_ordering;             /* resolved: 1 = internal locality ordering in use                             */
  int32_t halo_exchange;             /* resolved: 1 = point-to-point halo exchange, 0 = all-gather / single rank    */
  int32_t halo_overlap;              /* resolved                                                                    */
  int64_t halo_send_rows, halo_recv_rows;
  int64_t device_bytes;              /* HBM allocated by the handle                                                 */
  double  host_enqueue_us_per_pcg_iter; /* host time spent in launch calls per enqueued PCG iteration so far (no waiting);
                                        with graph replay ~0, eager multi-rank loops: launches + collective calls          */
  int32_t pcg_graph_replay;          /* 1 = the PCG slices are replayed from a captured hipGraph                    */
  int32_t linear_solver;             /* resolved: 1 = PCG, 2 = direct (chain + low rank)                            */
  int32_t direct_rank;               /* order of the direct solve's dense capacitance matrix (3 x edges outside the chain) */
  int32_t direct_fallbacks;          /* LM iterations whose direct solve gave no usable step and were redone by PCG  */
  int32_t direct_switched_at;        /* auto, rank above 2048: the LM iteration after which the direct solve took over from PCG
                                        (its PCG solve cost more than a direct solve of this rank does), 0 = it has not       */
  int32_t pcg_coarse_poses;          /* resolved: poses per aggregate of the second preconditioner level, 0 = one level   */
  int32_t pcg_coarse_rank;           /* order of its dense coarse matrix                                               */
  int32_t pcg_single_reduction;      /* 1 = the PCG loop with ONE reduction point per iteration (Chronopoulos-Gear recurrences:
                                        world > 1, pcg_rtol >= 1e-6, chain preconditioner), 0 = the textbook two-reduction loop */
  int32_t _pad;
} pgo_handle_info;
int pgo_get_info(const pgo_t* h, pgo_handle_info* out);                           /* [host] */

int pgo_get_poses(pgo_t* h, double* out_xyt /* N x 3 */);                         /* [gpu] */
/* METHOD 2: current switch per edge in the caller's edge order (1.0 for odometry edges); optionally also
 * d e / d s (E x 3, after the Huber corrector) of the latest Jacobian evaluation.  world == 1.             */
int pgo_get_switches(pgo_t* h, double* switches /* E */, double* js_or_null /* E x 3 */);   /* [gpu] */
int pgo_set_poses(pgo_t* h, const double* poses_xyt);                             /* [gpu] */

/* ------------------------------------------------ kernel-level entry points
 * Used by the parity tests and by bench.py's roofline leg: each launches exactly
 * one kind of kernel `reps` times on the handle's stream, brackets the launches
 * with HIP events on that stream and returns the average milliseconds.          */
typedef struct pgo_kernel_stats {
  double ms_avg;              /* average launch duration (HIP events)            */
  double algorithmic_bytes;   /* bytes one launch must move    */
  int64_t units;              /* edges (K1/K2) or blocks (K3) per launch         */
} pgo_kernel_stats;
int pgo_bench_eval(pgo_t* h, int reps, int with_jacobian, pgo_kernel_stats* out); /* [gpu] K1 */
int pgo_bench_assemble(pgo_t* h, int reps, pgo_kernel_stats* out);                /* [gpu] K2 */
int pgo_bench_spmv(pgo_t* h, int reps, pgo_kernel_stats* out);                    /* [gpu] K3 */
int pgo_debug_precond(pgo_t* h, const double* r_3n, double* z_3n);               /* [gpu] z = M^-1 r, current preconditioner */
int pgo_bench_precond(pgo_t* h, int reps, pgo_kernel_stats* out);                 /* [gpu] z = M^-1 b as the PCG start-up kernel */
/* y = (J'J + D'D) x in the scaled space at the current linearisation, with the
 * current LM diagonal; x,y: 3N doubles (world == 1).  For SpMV parity tests.    */
int pgo_debug_spmv(pgo_t* h, const double* x, double* y);                         /* [gpu] */
/* normal-equation pieces at the current point, caller's pose order (world == 1):
 * g: 3N gradient J'r (unscaled), hdiag: N x 9 diagonal 3x3 blocks of J'J         */
int pgo_debug_normal_eq(pgo_t* h, double* g_or_null, double* hdiag_or_null);      /* [gpu] */
/* Test hooks -- for tests/ only, not part of the drop-in surface.  Process-wide knobs read by pgo_create* (handles
 * created afterwards); value < 0 restores the library's default.  The library reads NO environment variable other than
 * PGO_FORCE_COLLECTIVES (1 = issue the collectives at world == 1 too, where they are identities) and
 * PGO_GRAPH_COLLECTIVES (0 = never capture collectives into the PCG hipGraph, 1 = all-reduce / all-gather, 2 = also the
 * point-to-point exchange), and never lets the environment override a pgo_options field.
 *   "spmv_pipe"          0 = K3 as k_spmv_t; 2 = K3 as the software-pipelined k_spmv_p even where k_spmv_1 (one row tile
 *                        per workgroup, graphs with more than 4096 tiles) is the default (all three must agree)
 *   "fused_p"            0 = small graphs keep the three-launch PCG loop (no direction update inside the SpMV)
 *   "direct_fail_at"     k = the direct solve of LM iteration k returns NaNs (exercises the PCG redo)
 *   "direct_setup_fail"  1 = setting up the direct solver fails with PGO_ERR_NOMEM after its first allocations
 *   "single_reduction"   1 / 0 = force the one-reduction (Chronopoulos-Gear) PCG loop on / off (default: on for
 *                        world > 1 in the inexact mode, pcg_rtol >= 1e-6)
 *   "verify_residual"    1 = pcg_rel_residual of the iteration records is the TRUE |b - A y| / |b| of each PCG solve (one
 *                        more product per solve) instead of the recurrence residual the loop stopped on
 *   "shm_timeout_s"      seconds a rank of the shm TEST communicator waits at a barrier before it gives up (default 120)
 *   "pad_tiles"          0 = large graphs keep the dense incidence layout (default: every row tile padded to 256 incidence
 *                        slots of its own, so that K3 finds a tile's blocks from its number alone; same results);
 *                        1 = that layout and its product kernel (k_spmv_1) on a graph of any size
 * Unknown name: PGO_ERR_INVALID_ARG.                                                                              */
int pgo_debug_set_knob(const char* name, long long value);                        /* [host] */
/* sharding plan of a graph over `world` ranks: for rank r, rows [lo, hi) and the
 * number of local edges / cut edges.  rows per rank = ceil(N / world) rounded up to a
 * multiple of row_align (the solver passes its preconditioner block size, see
 * pgo_options.pcg_block_poses; 1 = plain ceil).  Pure host logic.                */
int pgo_shard_plan(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib,
                   int world, int rank, int row_align, int32_t* lo, int32_t* hi,
                   int32_t* n_local_edges, int32_t* n_cut_edges);                 /* [host] */
/* Locality ordering of the poses, the permutation the solver applies internally when pose_ordering = 1:
 * perm[i] = new position of pose i.  Segments of `segment` consecutive poses stay contiguous and in order (the
 * odometry chain and the preconditioner's pose blocks survive; `segment` should be a multiple of the block size);
 * the segments are reordered by reverse Cuthill-McKee on the graph of SUPPORTED loop edges -- (a, b) is supported
 * when some edge joins {a-1, a, a+1} x {b-1, b, b+1} other than itself, which keeps the mesh of true revisits and drops
 * isolated random loops; the last (short) segment stays last.  Pure host logic.                                     */
int pgo_pose_order(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib, int32_t segment,
                   int32_t* perm);                                                /* [host] */
/* halo of that plan: send_rows[s] = how many of rank's rows peer s references, recv_rows[s] = how many of
 * peer s's rows rank references (arrays of `world` entries; the own-rank entries are 0).  Pure host logic.  */
int pgo_shard_halo(int32_t n_poses, int32_t n_edges, const int32_t* ia, const int32_t* ib,
                   int world, int rank, int row_align, int64_t* send_rows, int64_t* recv_rows); /* [host] */

#ifdef __cplusplus
}
#endif
#endif /* PGO_H_ */
