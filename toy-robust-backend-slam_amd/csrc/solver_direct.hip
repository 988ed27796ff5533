// The direct linear solve for small chain-like graphs (kernels: direct.hip.h; choice and buffers: solver_create.hip).
#include "solver_handle.hip.h"

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
  hipLaunchKernelGGL(dev::k_dlr_setup<>, dim3((n + dl_m + 255) / 256), dim3(256), 0, stream, A);
  PGOC(check_launch("k_dlr_setup"));
  hipLaunchKernelGGL(dev::k_dlr_factor<>, dim3(dl_nsep + 1), dim3(64), 0, stream, (const double*)dl_trec, n, dl_fac, A);
  PGOC(check_launch("k_dlr_factor"));
  hipLaunchKernelGGL(dev::k_dlr_prefix<>, dim3(1), dim3(128), 0, stream, (const double*)dl_fac, n, dl_nseg, dl_seglen, dl_pre);
  PGOC(check_launch("k_dlr_prefix"));
  const bool one_launch = dl_refine > 0 && dl_pre2 != nullptr;   // the refinement's single column: k_dlr_solve1
  if (one_launch) {
    hipLaunchKernelGGL(dev::k_dlr_prefix<>, dim3(1), dim3(512), 0, stream, (const double*)dl_fac, n, dl_nseg2, dl_seglen2, dl_pre2);
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
    hipLaunchKernelGGL(dev::k_dlr_fwd<>, grid, dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_mid<>, grid, dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_fix<>, grid, dim3(256), 0, stream, Q);
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
    hipLaunchKernelGGL(dev::k_dlr_sep_w<>, dim3((ncols + 255) / 256), dim3(256), 0, stream, Q);
    hipLaunchKernelGGL(dev::k_dlr_sep_apply<>, dim3((ncols + 255) / 256, (3 * n + 63) / 64), dim3(256), 0, stream, Q);
    return check_launch("k_dlr_sep_w / _apply");
  };
  if (dl_nsep > 0) {
    SA.X = dl_Z;
    SA.ld = dl_ld;
    SA.ncols = K + 1;
    hipLaunchKernelGGL(dev::k_dlr_sep_system<>, dim3(1), dim3(256), 0, stream, SA);
    PGOC(check_launch("k_dlr_sep_system"));
  }
  PGOC(separator_fix(dl_Z, dl_ld, K + 1));
  hipLaunchKernelGGL(dev::k_dlr_cap<>, dim3((std::max(Kp, K + 1) + 255) / 256, Kp), dim3(256), 0, stream, A);
  PGOC(check_launch("k_dlr_cap"));
  for (int kb = 0; kb < nb; ++kb) {
    hipLaunchKernelGGL(dev::k_chol_panel<>, dim3(std::max(1, nb - 1)), dim3(dev::CHOL_THREADS), dev::CHOL_LDS_BYTES, stream, dl_cap, dl_nm, dl_dwork, Kp, nb, kb);
    PGOC(check_launch("k_chol_panel"));
  }
  auto capacitance_solve = [&]() -> int {  // cvec <- (L L')^-1 cvec = N' (N cvec)
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)dl_nm, Kp, nb, (const double*)dl_cvec, dl_cy, 0);
    hipLaunchKernelGGL(dev::k_tri_apply<>, dim3(nb), dim3(256), 0, stream, (const double*)dl_nm, Kp, nb, (const double*)dl_cy, dl_cvec, 1);
    return check_launch("k_tri_apply");
  };
  PGOC(capacitance_solve());
  hipLaunchKernelGGL(dev::k_dlr_combine<>, dim3((3 * n + 3) / 4), dim3(256), 0, stream, (const double*)dl_Z, dl_ld, K, (const double*)dl_cvec,
                     (const double*)dl_Z, dl_ld, K, 3 * n, y, 0);
  PGOC(check_launch("k_dlr_combine"));
  auto residual_product = [&]() -> int {  // ap = (H + D'D) y
    hipLaunchKernelGGL(dev::k_scatter_owned<>, dim3(g_flat), dim3(dev::WG), 0, stream, S.n_loc, S.lo, y, p_full);
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
      hipLaunchKernelGGL(dev::k_dlr_solve1<>, dim3(1), dim3(256), 0, stream, Q);
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
    hipLaunchKernelGGL(dev::k_dlr_vdot<>, dim3((Kp + 255) / 256), dim3(256), 0, stream, A, (const double*)dl_x1, xld, 0, dl_cvec);
    PGOC(check_launch("k_dlr_vdot"));
    PGOC(capacitance_solve());
    hipLaunchKernelGGL(dev::k_dlr_combine<>, dim3((3 * n + 3) / 4), dim3(256), 0, stream, (const double*)dl_Z, dl_ld, K, (const double*)dl_cvec,
                       (const double*)dl_x1, xld, 0, 3 * n, y, 1);
    PGOC(check_launch("k_dlr_combine"));
  }
  PGOC(residual_product());
  hipLaunchKernelGGL(dev::k_dlr_resid<>, dim3((3 * n + 255) / 256), dim3(256), 0, stream, (int64_t)3 * n, (const double*)gs, (const double*)ap, r);
  PGOC(check_launch("k_dlr_resid"));
  hipLaunchKernelGGL(dev::k_dot<>, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * n, (const double*)r, (const double*)r, part[2]);
  hipLaunchKernelGGL(dev::k_dot<>, dim3(g_flat), dim3(dev::WG), 0, stream, (int64_t)3 * n, (const double*)gs, (const double*)gs, part[4]);
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

