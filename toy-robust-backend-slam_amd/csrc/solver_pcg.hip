// The linear solve of an LM iteration by preconditioned CG on the Jacobi-scaled normal equations (north_star: "block-Jacobi-
// preconditioned CG whose SpMV is a CDNA4 HIP kernel"): the PCG loops (textbook, fused direction update for small graphs,
// single reduction for several ranks), the preconditioners' per-iteration set-up, the coarse level's factorisation and apply.
#include "solver_handle.hip.h"

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
  hipLaunchKernelGGL(dev::k_coarse_basis<>, dim3((co_nagg + 3) / 4), dim3(256), 0, stream, A);
  PGOC(check_launch("k_coarse_basis"));
  HIPC(hipMemsetAsync(co_cap, 0, (size_t)co_Kp * co_Kp * sizeof(double), stream));
  HIPC(hipMemsetAsync(co_dwork, 0, (size_t)(co_Kp / 32) * 1024 * sizeof(double), stream));
  hipLaunchKernelGGL(dev::k_coarse_assemble<>, dim3((co_ncb + 3) / 4), dim3(256), 0, stream, A);
  PGOC(check_launch("k_coarse_assemble"));
  if (co_Kp > co_K) {
    hipLaunchKernelGGL(dev::k_coarse_pad<>, dim3(1), dim3(32), 0, stream, co_cap, co_dwork, co_K, co_Kp);
    PGOC(check_launch("k_coarse_pad"));
  }
  const int nb = co_Kp / 32;
  for (int kb = 0; kb < nb; ++kb) {
    hipLaunchKernelGGL(dev::k_chol_panel<>, dim3(std::max(1, nb - 1)), dim3(dev::CHOL_THREADS), dev::CHOL_LDS_BYTES, stream, co_cap, co_nm, co_dwork, co_Kp, nb, kb);
    PGOC(check_launch("k_chol_panel (coarse level)"));
  }
  // usable?  a probe through the factor: x = A_c^-1 1 must be finite (a pivot lost to rounding leaves NaNs behind it)
  hipLaunchKernelGGL(dev::k_fill<>, dim3((co_Kp + 255) / 256), dim3(256), 0, stream, co_rc, (int64_t)co_Kp, 1.0);
  if (co_ainv) {
    hipLaunchKernelGGL(dev::k_coarse_ainv<>, dim3((co_Kp + 255) / 256, co_Kp), dim3(256), 0, stream, (const double*)co_nm, co_Kp, co_ainv);
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)co_ainv, co_Kp, nb, (const double*)co_rc, co_ec, 0);
  } else {
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_rc, co_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_cy, co_ec, 1);
  }
  hipLaunchKernelGGL(dev::k_coarse_check<>, dim3(1), dim3(256), 0, stream, (const double*)co_ec, co_Kp, co_ok);
  // (the restriction writes the K coarse unknowns only: the padding entries K .. Kp-1 of r_c must read 0 again)
  hipLaunchKernelGGL(dev::k_fill<>, dim3((co_Kp + 255) / 256), dim3(256), 0, stream, co_rc, (int64_t)co_Kp, 0.0);
  return check_launch("coarse level probe");
}

int pgo_handle::coarse_solve(double* dot_part, const int32_t* done) {
  const int nb = co_Kp / 32;
  hipLaunchKernelGGL(dev::k_coarse_restrict<>, dim3((co_nagg + 3) / 4), dim3(256), 0, stream, (int)S.n_loc, co_agg, co_nagg, (const double*)co_pb,
                     (const double*)r, co_rc, done);
  if (co_ainv) {
    hipLaunchKernelGGL(dev::k_coarse_matvec<>, dim3(co_ndot), dim3(256), 0, stream, (const double*)co_ainv, co_Kp, (const double*)co_rc, co_ec,
                       dot_part, (const int32_t*)co_ok, done);
  } else {
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_rc, co_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)co_nm, co_Kp, nb, (const double*)co_cy, co_ec, 1);
    hipLaunchKernelGGL(dev::k_coarse_dot<>, dim3(co_ndot), dim3(256), 0, stream, co_Kp, (const double*)co_rc, co_ec, dot_part,
                       (const int32_t*)co_ok, done);
  }
  return check_launch("coarse level solve");
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
    hipLaunchKernelGGL(dev::k_pcg_solo<>, dim3(1), dim3(dev::SOLO_WG), 0, stream, A);
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
  else if (grouped) hipLaunchKernelGGL(dev::k_cg_init_g<>, dim3(g_u1), dim3(dev::WG), 0, stream, V, GP, (const double*)gs, part[0], part[1]);
  else hipLaunchKernelGGL(dev::k_cg_init<>, dim3(g_u1), dim3(dev::WG), 0, stream, V, gs, part[0], part[1]);
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
    hipLaunchKernelGGL(dev::k_cg_sr_scal<>, dim3(1), dim3(1), 0, stream, st, (const double*)(scal + 4), opt.pcg_rtol, first);
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
      hipLaunchKernelGGL(dev::k_coarse_prolong<>, dim3((unsigned)std::min<int64_t>((S.n_loc + 255) / 256, 512)), dim3(256), 0, stream, (int)S.n_loc,
                         co_agg, (const double*)co_pb, (const double*)co_ec, z, p_full + dev::PS * (int64_t)S.lo, (const int32_t*)co_ok);
      PGOC(check_launch("k_coarse_prolong"));
    }
    PGOC(reduce_to_scal({{part[0], n_rz0, 0}, {part[1], g_u1, 0}}, 4));
    hipLaunchKernelGGL(dev::k_cg_init_fin<>, dim3(1), dim3(1), 0, stream, st, scal + 4, opt.pcg_rtol);
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
    // p.Ap: every workgroup of the update kernel re-sums the product's partials itself -- up to 2048 of them; more (k_spmv_1:
    // one per tile) are folded to 16 first (k_fold_partials); several ranks: k_finalize + all-reduce
    const bool fold_pap = !multi && n_sp > 2048;
    const double* pap = multi ? scal + 6 : (fold_pap ? part[3] : part[0]);
    const int n_pap = multi ? 1 : (fold_pap ? 16 : n_sp);
    if (multi) PGOC(reduce_to_scal({{part[0], n_sp, 0}}, 6));
    else if (fold_pap) {
      hipLaunchKernelGGL(dev::k_fold_partials<>, dim3(16), dim3(dev::WG), 0, stream, (const double*)part[0], n_sp, part[3], (const int32_t*)&st->done);
      PGOC(check_launch("k_fold_partials"));
    }
    if (chained) launch_cg_update1_chain(Vi, par, pap, n_pap, part[1], part[2]);
    else if (grouped) hipLaunchKernelGGL(dev::k_cg_update1_g<>, dim3(g_u1), dim3(dev::WG), 0, stream, Vi, GP, par, pap, n_pap, part[1], part[2]);
    else hipLaunchKernelGGL(dev::k_cg_update1<>, dim3(g_u1), dim3(dev::WG), 0, stream, Vi, par, pap, n_pap, part[1], part[2]);
    PGOC(check_launch("k_cg_update1"));
    if (fused) return PGO_OK;  // its r.z / r.r partials are booked by the next SpMV, or by k_cg_book at the end of the slice
    if (use_coarse) {   // second level (single rank): e_c, its share of r.z as more partials, prolongation inside the direction update
      PGOC(coarse_solve(part[1] + g_u1, &st->done));
      hipLaunchKernelGGL(dev::k_cg_update2c<>, dim3(g_vec), dim3(dev::WG), 0, stream, V, par, (const double*)part[1], g_u1 + co_ndot,
                         (const double*)part[2], g_u1, co_agg, (const double*)co_pb, (const double*)co_ec);
      return check_launch("k_cg_update2c");
    }
    if (multi) {
      PGOC(reduce_to_scal({{part[1], g_u1, 0}, {part[2], g_u1, 0}}, 7));
      hipLaunchKernelGGL(dev::k_cg_update2<>, dim3(g_flat), dim3(dev::WG), 0, stream, V, par, scal + 7, 1, scal + 8, 1);
      PGOC(check_launch("k_cg_update2"));
      if (!overlap) PGOC(share_gather_vector(p_full));
    } else {
      hipLaunchKernelGGL(dev::k_cg_update2<>, dim3(g_flat), dim3(dev::WG), 0, stream, V, par, part[1], g_u1, part[2], g_u1);
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
        hipLaunchKernelGGL(dev::k_cg_book<>, dim3(1), dim3(dev::WG), 0, stream, st, (every - 1) & 1, (const double*)part[1], g_u1, (const double*)part[2], g_u1);
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
          hipLaunchKernelGGL(dev::k_cg_book<>, dim3(1), dim3(dev::WG), 0, stream, st, (it + chunk - 1) & 1, (const double*)part[1], g_u1, (const double*)part[2], g_u1);
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
  if (verify_residual) {
    // test hook: report |b - A y| / |b| of the solution instead of the recurrence residual the loop stopped on (one more
    // product; the recurrences of either loop drift from it on ill-conditioned systems)
    hipLaunchKernelGGL(dev::k_scatter_owned<>, dim3(g_flat), dim3(dev::WG), 0, stream, (int)S.n_loc, (int)S.lo, (const double*)V.y, p_full);
    PGOC(check_launch("k_scatter_owned"));
    PGOC(share_gather_vector(p_full));
    PGOC(spmv_enqueue(p_full, ap, part[0], 1, nullptr));
    hipLaunchKernelGGL(dev::k_residual_norm<>, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * S.n_loc, (const double*)gs, (const double*)ap, part[1], part[2]);
    PGOC(check_launch("k_residual_norm"));
    PGOC(reduce_to_scal({{part[1], g_flat, 0}, {part[2], g_flat, 0}}, 4));
    double h2[2] = {0.0, 0.0};
    HIPC(hipMemcpyAsync(h2, scal + 4, sizeof h2, hipMemcpyDeviceToHost, stream));
    PGOC(sync());
    *rel = (h2[1] > 0.0) ? std::sqrt(h2[0] / h2[1]) : 0.0;
  }
  return PGO_OK;
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
    hipLaunchKernelGGL(dev::k_prepare_groups<>, dim3(grp_prep_grid), dim3(dev::WG), grp_lds, stream, GA);
    PGOC(check_launch("k_prepare_groups"));
  }
  if (chain_len) PGOC(factor_chain());
  if (use_coarse) PGOC(coarse_factor());
  return PGO_OK;
}

// block LDL' of the chain preconditioner's segments (the records are complete: C part from k_assemble, M part from k_prepare)
int pgo_handle::factor_chain() {
  const int n_seg = (S.n_loc + chain_len - 1) / chain_len;
  // One THREAD per segment runs the recurrence (64 .. 256 dependent steps), 64 segments per wavefront: 244 wavefronts at 1M
  // poses.  Fewer segments per wavefront (16: four wavefronts per compute unit, each with its own queue of outstanding loads)
  // were measured SLOWER -- 233 us against 123 us: the same 128-byte-per-lane record loads then take four times as many
  // wave instructions through the address coalescer, and 174 instead of 122 MB are written.
  const int spw = 64;
  if (n_seg == 0) return PGO_OK;   // a rank that owns no rows
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
  hipLaunchKernelGGL(dev::k_prepare<>, dim3(g_rows), dim3(dev::WG), 0, stream, hd, (const double*)diag_full, S.n_loc, S.lo, fixed_internal, radius,
                     opt.min_lm_diagonal, opt.max_lm_diagonal, d2, minv, (const uint8_t*)fixed_mask, (const int32_t*)prob_of_256,
                     (const double*)prob_radius, chain_len ? chain_c : (double*)nullptr, hdd);
  PGOC(check_launch("k_prepare"));
  if (!direct) PGOC(prepare_preconditioner());   // (the direct solve does not need it; its PCG fallback sets it up on demand)
  return PGO_OK;
}

