// pgo_handle::create -- replaces the problem assembly of DCS-ceres/main.cpp:66-68,95-153: shard structure (structure.cpp) ->
// device buffers, grids, and the choices made once per handle: preconditioner family, direct solve (direct_setup), second
// preconditioner level (coarse_setup).
#include "solver_handle.hip.h"

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
  // Graphs large enough for the one-tile-per-workgroup product kernel (k_spmv_1, below) get the padded-slot layout: every
  // tile's incidences at TILE_INC t (structure.cpp, pad_tiles_to_slots).  Test hook "pad_tiles" = 0 keeps the dense layout.
  int one_tile_min = 4096;
  if (const char* om = PGO_EXP_ENV("PGO_ONE_TILE_MIN")) one_tile_min = atoi(om);
  if (knob("pad_tiles") == 1) one_tile_min = 0;   // (test hook: the large-graph layout and product kernel on a graph of any size)
  if (!batch_mode && S.n_tiles() > one_tile_min && knob("pad_tiles") != 0) (void)pgo::pad_tiles_to_slots(&S);
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
  g_asm = up8(std::min(std::max(1, S.n_tiles()), 1 << 20));
  part_cap = std::max(std::max(g_edge, 2048), up8(std::max(1, S.n_tiles()))) + 8 + 512;   // (k_spmv_1: one dot partial per tile)   // (+ the coarse level's dot partials behind the one-level r.z partials)
  for (int k = 0; k < N_PART; ++k) PGOC(dalloc(&part[k], part_cap));
  PGOC(dalloc(&fold_buf, 6 * 16));

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
        // Large graphs: one tile per workgroup (k_spmv_1) -- 151 us against the pipelined form's 164-166 us at 1M poses; test
        // hook "spmv_pipe" = 2 keeps k_spmv_p there.  Up to 4096 tiles the persistent forms stay: fewer partials, no
        // k_fold_partials launch in a latency-bound iteration (100k poses, 3.2k tiles: 247 vs 237 GN it/s; 60k: 369 vs 329).
        if (S.n_tiles() > one_tile_min && knob("spmv_pipe") != 2 && up8(S.n_tiles()) + 8 + 512 <= part_cap) {
          spmv_one_tile = true;
          g_spmv = up8(S.n_tiles());
        }
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
      HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_prepare_groups<>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)grp_lds));
  } else {
    grp_B = 1;
  }
  if (chain_len) {   // (also on a rank that owns no rows: every rank must resolve to the same PCG loop and the same collectives)
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
    g_chain = (int)std::max<int64_t>(1, std::min<int64_t>((NL + 4 * dev::CHAIN_TILE - 1) / (4 * dev::CHAIN_TILE), 2048));
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
      g_chain = (int)std::max<int64_t>(1, std::min<int64_t>((n_wt + chain_nw - 1) / chain_nw, chain_nw == 1 ? 2048 : cap));
    }
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
    use_sr = chain_len > 0 && chain_chunk > 0 && !solo && !fused_p && !batch_mode && opt.pcg_rtol >= 1e-6 &&
             (kn == 1 || (kn != 0 && (world > 1 || force_collectives)));
    if (use_sr) PGOC(dalloc(&sr_s, 3 * NL));
    verify_residual = knob("verify_residual") == 1;
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
    // auto (measured on MI355X, scripts/exp_coarse.py, GN it/s one level -> two levels):
    //   tight solves (pcg_rtol <= 1e-3) of graphs that stay on PCG -- M3500 METHOD 1 47 -> 156 (16-pose aggregates, coarse
    //   order 657; 32: 128, 64: 100), FRH 35 -> 199 (16), synthetic 10k at 1e-10 108 -> 137 (64; 16: 127), 100k at 1e-10
    //   3.7 -> 12.3 and at 1e-3 14.9 -> 40.5 (64, order 4689; 128 and more: no gain) -- 16 poses up to 8192 poses, else
    //   64, doubled until the coarse order fits the dense factorisation;
    //   loose solves (the inexact mode) stay on one level: measured both ways -- synthetic 10k at rtol 0.1, 64-pose aggregates:
    //   371 -> 771 over a long run at large radius, but 582 -> 530 over the first 50 LM iterations, where a solve needs few PCG
    //   iterations anyway and the coarse factorisation per LM iteration is pure overhead; 100k: 189 -> 80 .. 121.
    // ... and only while the caller left the one-level preconditioner to the library (like the direct solve's auto rule): an
    // explicit pcg_block_poses / pcg_chain_len is a request for exactly that preconditioner
    if (direct || NL < 512 || opt.pcg_chain_len != -1 || opt.pcg_block_poses != 0 || !(opt.pcg_rtol <= 1e-3)) return PGO_OK;
    want = NL <= 8192 ? 16 : 64;
    while (3 * ((NL + want - 1) / want) + 1 > COARSE_MAX_RANK) want *= 2;
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
  HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_chol_panel<>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::CHOL_LDS_BYTES));
  use_coarse = true;
  if (opt.linear_solver == 0) dl_possible = false;   // (ranks above the direct solve's cheap range: two-level PCG instead of the PCG / direct alternation)
  // the loops that fold launches together assume the one-level preconditioner: two-level solves take the plain three-kernel loop
  solo = false;
  fused_p = false;
  use_sr = false;
  return PGO_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Direct solve for small chain-like graphs (direct.hip.h).  opt.linear_solver: 0 = auto, 1 = PCG, 2 = direct.
// Auto picks it in the "exact" mode only (pcg_rtol <= 1e-8, where PCG stands in for the reference's
// SPARSE_NORMAL_CHOLESKY, main.cpp:154-163) and only while the caller left the preconditioner to the library; it needs
// one rank, METHOD 0 / 1, a constant pose, an edge between every pair of consecutive poses, and few enough other edges.

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
  HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(dev::k_chol_panel<>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev::CHOL_LDS_BYTES));
  PGOC(dalloc(&dl_cvec, dl_Kp));
  PGOC(dalloc(&dl_ksep, 18 * std::max(1, dl_nsep)));
  PGOC(dalloc(&dl_R, std::max(1, dl_nU * dl_nU)));
  PGOC(dalloc(&dl_Wm, (int64_t)std::max(1, dl_nU) * dl_ld));
  PGOC(sync());  // the host lists die with this scope
  direct = true;
  dl_ready = true;
  return PGO_OK;
}

