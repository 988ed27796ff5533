// The handle's launch helpers for the kernels several translation units need (K1, K2, K3, the reductions to scalars, the
// halo exchange, the chain-preconditioned PCG kernels): defined ONCE here, so that the other units do not carry copies of
// those kernels.
#include "solver_handle.hip.h"

int pgo_handle::reduce_to_scal(std::initializer_list<PartRef> parts, int first, bool allreduce_max) {
  dev::FinArgs F;
  memset(&F, 0, sizeof F);
  int k = 0;
  for (const PartRef& pr : parts) {
    F.part[k] = pr.p;
    F.n[k] = pr.n;
    F.is_max[k] = pr.is_max;
    if (!pr.is_max && pr.n > 4096 && fold_buf) {
      // k_finalize is ONE workgroup: tens of thousands of partials (k_spmv_1: one per tile) go through 16 workgroups first
      hipLaunchKernelGGL(dev::k_fold_partials<>, dim3(16), dim3(dev::WG), 0, stream, pr.p, pr.n, fold_buf + 16 * k, (const int32_t*)nullptr);
      PGOC(check_launch("k_fold_partials"));
      F.part[k] = fold_buf + 16 * k;
      F.n[k] = 16;
    }
    ++k;
  }
  F.count = k;
  F.out = scal + first;
  hipLaunchKernelGGL(dev::k_finalize<>, dim3(1), dim3(dev::WG), 0, stream, F);
  PGOC(check_launch("k_finalize"));
  if (multi_rank()) PGOC(comm->allreduce(scal + first, k, allreduce_max, stream));
  return PGO_OK;
}

int pgo_handle::share_gather_vector(double* full) {
  if (!multi_rank()) return PGO_OK;
  if (!use_halo) return allgather(full, dev::PS);
  const int64_t ns = (int64_t)S.halo_send_row.size(), nr = (int64_t)S.halo_recv_row.size();
  if (ns > 0) {
    hipLaunchKernelGGL(dev::k_pack_rows<>, dim3((unsigned)std::min<int64_t>((3 * ns + 255) / 256, 2048)), dim3(256), 0, stream, ns,
                       (const int32_t*)halo_send_rows, (const double*)full, halo_send_buf);
    PGOC(check_launch("k_pack_rows"));
  }
  PGOC(comm->exchange(halo_send_buf, halo_send_off3.data(), halo_recv_buf, halo_recv_off3.data(), stream));
  if (nr > 0) {
    hipLaunchKernelGGL(dev::k_unpack_rows<>, dim3((unsigned)std::min<int64_t>((3 * nr + 255) / 256, 2048)), dim3(256), 0, stream, nr,
                       (const int32_t*)halo_recv_rows, (const double*)halo_recv_buf, full);
    PGOC(check_launch("k_unpack_rows"));
  }
  return PGO_OK;
}

void pgo_handle::launch_eval(const double* x, const double* sw_vals, int apply_loss, bool with_jac) {
  dev::EdgeArgs A = edge_args(x, sw_vals, apply_loss);
  if (info_mode) {
    if (with_jac) hipLaunchKernelGGL((dev::k_edge_eval<true, true>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
    else hipLaunchKernelGGL((dev::k_edge_eval<false, true>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
  } else {
    if (with_jac) hipLaunchKernelGGL((dev::k_edge_eval<true, false>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
    else hipLaunchKernelGGL((dev::k_edge_eval<false, false>), dim3(g_edge), dim3(dev::WG), 0, stream, A, jr, part[5], bad);
  }
}

int pgo_handle::eval_enqueue(const double* x, const double* sw_vals, int apply_loss, bool with_jac, int slot) {
  HIPC(hipMemsetAsync(bad, 0, sizeof(int), stream));
  launch_eval(x, sw_vals, apply_loss, with_jac);
  PGOC(check_launch("k_edge_eval"));
  // the flag rides along as a "partial array" of length 1 after conversion to double
  hipLaunchKernelGGL(dev::k_flag_to_double<>, dim3(1), dim3(1), 0, stream, bad, part[4]);
  return reduce_to_scal({{part[5], g_edge, 0}, {part[4], 1, 0}}, slot);
}

int pgo_handle::assemble_enqueue() {
  if (S.n_tiles() == 0) return PGO_OK;
  if (has_sw) hipLaunchKernelGGL((dev::k_assemble<true, false>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
  else if (info_mode) hipLaunchKernelGGL((dev::k_assemble<false, true>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
  else hipLaunchKernelGGL((dev::k_assemble<false, false>), dim3(g_asm), dim3(dev::WG), 0, stream, asm_args());
  PGOC(check_launch("k_assemble"));
  if (chain_len && n_chain_dup > 0) {
    hipLaunchKernelGGL(dev::k_chain_dupfix<>, dim3((n_chain_dup + 63) / 64), dim3(64), 0, stream, (const int32_t*)chain_dup_rows, n_chain_dup,
                       (const int32_t*)inc_ptr, (const int32_t*)inc_col, (const double*)hoff, S.lo, chain_c);
    PGOC(check_launch("k_chain_dupfix"));
  }
  return PGO_OK;
}

int pgo_handle::spmv_enqueue(const double* p, double* yout, double* dot_part, int with_d2, const int32_t* done) {
  dev::SpmvArgs A = spmv_args(p, yout, dot_part, with_d2, done);
  switch (spmv_ablate) {
#ifdef PGO_EXPERIMENTS
    case 1: hipLaunchKernelGGL(dev::k_spmv_t<1>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
    case 2: hipLaunchKernelGGL(dev::k_spmv_t<2>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
    case 3: hipLaunchKernelGGL(dev::k_spmv_t<3>, dim3(g_spmv), dim3(dev::WG), 0, stream, A); break;
#endif
    default:
      if (spmv_pipe && spmv_one_tile && S.padded) hipLaunchKernelGGL(dev::k_spmv_1<true>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
      else if (spmv_pipe && spmv_one_tile) hipLaunchKernelGGL(dev::k_spmv_1<false>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
      else if (spmv_pipe) hipLaunchKernelGGL(dev::k_spmv_p<dev::PS>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
      else hipLaunchKernelGGL(dev::k_spmv_t<0>, dim3(g_spmv), dim3(dev::WG), 0, stream, A);
  }
  return check_launch("k_spmv");
}

int pgo_handle::spmv_with_halo(double* full, double* yout, double* dot_part, const int32_t* done, int* n_part) {
  if (!overlap) {
    PGOC(share_gather_vector(full));
    *n_part = g_spmv;
    return spmv_enqueue(full, yout, dot_part, 1, done);
  }
  const int64_t ns = (int64_t)S.halo_send_row.size(), nr = (int64_t)S.halo_recv_row.size();
  if (ns > 0) {
    hipLaunchKernelGGL(dev::k_pack_rows<>, dim3((unsigned)std::min<int64_t>((3 * ns + 255) / 256, 2048)), dim3(256), 0, stream, ns,
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
    hipLaunchKernelGGL(dev::k_unpack_rows<>, dim3((unsigned)std::min<int64_t>((3 * nr + 255) / 256, 2048)), dim3(256), 0, comm_stream, nr,
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
    hipLaunchKernelGGL(dev::k_spmv_remote<>, dim3(g_rr), dim3(dev::WG), 0, stream, R);
    PGOC(check_launch("k_spmv_remote"));
    used += g_rr;
  }
  *n_part = used;
  return PGO_OK;
}

void pgo_handle::launch_cg_init_chain(const double* b, double* part_rz, double* part_bb) {
  const dev::CgVec V = cg_vec();
  const dev::ChainPre CP = chain_pre();
  if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
  else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_init_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
  else if (chain_chunk == 4 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
  else if (chain_chunk == 4) hipLaunchKernelGGL((dev::k_cg_init_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, b, part_rz, part_bb);
  else hipLaunchKernelGGL(dev::k_cg_init_c<>, dim3(g_chain), dim3(dev::WG), 0, stream, V, CP, b, part_rz, part_bb);
}

void pgo_handle::launch_cg_sr_chain(const dev::CgVec& V, double* part_gamma, double* part_rr) {
  const dev::ChainPre CP = chain_pre();
  if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_sr_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
  else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_sr_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
  else if (chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_sr_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
  else hipLaunchKernelGGL((dev::k_cg_sr_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, sr_s, chain_steps, chain_scan, part_gamma, part_rr);
}

void pgo_handle::launch_cg_update1_chain(const dev::CgVec& V, int par, const double* pap, int n_pap, double* part_rz, double* part_rr) {
  const dev::ChainPre CP = chain_pre();
  if (chain_chunk == 2 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<2, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
  else if (chain_chunk == 2) hipLaunchKernelGGL((dev::k_cg_update1_cl<2, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
  else if (chain_chunk == 4 && chain_nw == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<4, 4>), dim3(g_chain), dim3(256), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
  else if (chain_chunk == 4) hipLaunchKernelGGL((dev::k_cg_update1_cl<4, 1>), dim3(g_chain), dim3(64), 0, stream, V, CP, chain_steps, chain_scan, par, pap, n_pap, part_rz, part_rr);
  else hipLaunchKernelGGL(dev::k_cg_update1_c<>, dim3(g_chain), dim3(dev::WG), 0, stream, V, CP, par, pap, n_pap, part_rz, part_rr);
}
