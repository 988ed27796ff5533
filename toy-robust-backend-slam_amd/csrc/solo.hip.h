// One workgroup solves one problem: the WHOLE block-Jacobi PCG solve of an LM iteration in a single launch.
//
// Why: on the reference's own datasets (INTEL 1228 poses, MIT 808, CSAIL, FR079; BASELINE configs[1]-[2]) a PCG iteration
// of the three-kernel loop costs 28-30 us -- 4.8 us SpMV + 18-20 us chain apply + 4.7 us vector update, each of them a
// handful of dependent L2 round trips on 5-13 tiles, nothing to overlap with -- and an exact LM iteration needs ~300 of
// them.  Here the iteration's phases are separated by workgroup barriers instead of kernel boundaries, the scalars
// (alpha, beta, convergence) live in registers / LDS, every phase issues all of its loads before the first use, and the
// 256-pose chain segments of these graphs run the recurrence over the lanes as a 6-level scan.
// The same kernel is the engine of the batched solver (pgo_batch_*, SURVEY 8 f-4: the reference's layer managers issue
// thousands of 1-2-iteration solves of INTEL-sized problems, src/simple_layer_manager.cpp:457-622,
// src/layer_manager.cpp:137-179,602-654): grid = number of problems, one workgroup each, per-problem state on the device.
//
// Arithmetic: the same operator, preconditioner and update order as k_spmv / k_cg_update1_cl / k_cg_update2; only the
// association of the dot-product sums differs (per-workgroup instead of per-tile partials).
#pragma once
#include "kernels.hip.h"

namespace pgo {
namespace dev {

constexpr int SOLO_WG = 512;              // 8 waves: two 256-lane halves for the SpMV, 8 chain tiles at a time
constexpr int SOLO_U = 4;                 // row tiles per half per SpMV round
constexpr int SOLO_TILES = 2 * SOLO_U;    // row tiles in flight per round
constexpr int SOLO_CH = 4;                // chain layout: 256-row wave tiles

struct SoloProb {
  int32_t row0, nrows;      // local rows [row0, row0 + nrows); row0 is a multiple of 256
  int32_t tile0, ntiles;    // its row tiles in tile_desc
  int32_t active;           // 0: nothing to do (a terminated problem of a batch)
  int32_t max_it;
  double rtol;
};
struct SoloOut {
  double rz, bb, rr;               // PCG state at exit
  double ydotg, yHy, step2;        // epilogue: y.g, y.(H y) (without the LM diagonal), |S y|^2
  int32_t iters, done;
};
struct SoloArgs {
  SpmvArgs A;                      // matrix; A.p = the gather vector, A.y = A p
  CgVec V;
  ChainPre C;                      // chain factors in the 256-row tile layout (chunk 4); C.cw == nullptr: 3x3 block-Jacobi (V.minv)
  int32_t chain_steps, scan_levels;
  const double* b;                 // right-hand side (scaled gradient)
  const SoloProb* prob;
  SoloOut* out;
  const double* x;                 // epilogue inputs: current poses, Jacobi scales (global pose indexing)
  const double* scale;
  double* cand;                    // candidate poses out; nullptr = no epilogue
};

// sum over the 512 threads, broadcast.  sh: >= 9 doubles
__device__ __forceinline__ double solo_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[8] = ((sh[0] + sh[1]) + (sh[2] + sh[3])) + ((sh[4] + sh[5]) + (sh[6] + sh[7]));
  __syncthreads();
  return sh[8];
}

// yout[rows of the problem] = (H (+ D'D)) pvec ;  returns  sum_rows pvec_row . yout_row
template <bool WITH_D2>
__device__ __forceinline__ double solo_spmv(const SpmvArgs& A, const SoloProb& P, const double* __restrict__ pvec,
                                            double* __restrict__ yout, double (*scr)[3][256], double* red) {
  const int tid = threadIdx.x, half = tid >> 8, lt = tid & 255;
  const int64_t n = A.n_loc;
  double dot = 0.0;
  for (int base = 0; base < P.ntiles; base += SOLO_TILES) {
    // ---- every load of the round first: SOLO_U tiles per lane
    int4 d[SOLO_U];
    double pr[SOLO_U][3];           // staged products
    bool on[SOLO_U];
    int ra[SOLO_U], rrow[SOLO_U], rlo[SOLO_U], rhi[SOLO_U];
    double h0[SOLO_U], h1[SOLO_U], h2[SOLO_U], dd[SOLO_U], q0[SOLO_U], q1[SOLO_U], q2[SOLO_U];
#pragma unroll
    for (int u = 0; u < SOLO_U; ++u) {
      const int t = base + half * SOLO_U + u;
      d[u] = (t < P.ntiles) ? A.tile_desc[P.tile0 + t] : make_int4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SOLO_U; ++u) {
      const int r0 = d[u].x, nrows = d[u].y, qb = d[u].z, nq = d[u].w;
      on[u] = lt < nq;
      pr[u][0] = pr[u][1] = pr[u][2] = 0.0;
      if (on[u]) {
        const int q = qb + lt;
        const int64_t col = A.inc_col[q];
        double p0, p1, p2, h[9];
        gather3(pvec, col, p0, p1, p2);
        hoff_load(A.hoff, q, h);
        pr[u][0] = h[0] * p0 + h[1] * p1 + h[2] * p2;
        pr[u][1] = h[3] * p0 + h[4] * p1 + h[5] * p2;
        pr[u][2] = h[6] * p0 + h[7] * p1 + h[8] * p2;
      }
      ra[u] = -1;
      h0[u] = h1[u] = h2[u] = dd[u] = q0[u] = q1[u] = q2[u] = 0.0;
      rrow[u] = rlo[u] = rhi[u] = 0;
      if (lt < nrows * 3) {
        const int a = lt / nrows, row = r0 + (lt - a * nrows);
        ra[u] = a;
        rrow[u] = row;
        rlo[u] = A.inc_ptr[row] - qb;
        rhi[u] = A.inc_ptr[row + 1] - qb;
        const int i1 = (a == 0) ? 1 : (a == 1 ? 3 : 4), i2 = (a == 2) ? 5 : (a == 1 ? 4 : 2);
        h0[u] = A.hd[(int64_t)a * n + row];
        h1[u] = A.hd[(int64_t)i1 * n + row];
        h2[u] = A.hd[(int64_t)i2 * n + row];
        if (WITH_D2) dd[u] = A.d2[3 * (int64_t)row + a];
        const double* pp = pvec + PS * (int64_t)(A.lo + row);
        q0[u] = pp[0];
        q1[u] = pp[1];
        q2[u] = pp[2];
      }
    }
    // ---- stage, barrier, segmented row sums
#pragma unroll
    for (int u = 0; u < SOLO_U; ++u)
      if (on[u]) {
        double(*sc)[256] = scr[half * SOLO_U + u];
        sc[0][lt] = pr[u][0];
        sc[1][lt] = pr[u][1];
        sc[2][lt] = pr[u][2];
      }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SOLO_U; ++u) {
      double(*sc)[256] = scr[half * SOLO_U + u];
      if (ra[u] >= 0) {
        const int a = ra[u];
        double s = 0.0;
        for (int j = rlo[u]; j < rhi[u]; ++j) s += sc[a][j];
        const double pa = (a == 0) ? q0[u] : (a == 1 ? q1[u] : q2[u]);
        s += h0[u] * q0[u] + h1[u] * q1[u] + h2[u] * q2[u] + dd[u] * pa;
        yout[3 * (int64_t)rrow[u] + a] = s;
        dot += pa * s;
      }
      const int r0 = d[u].x, nrows = d[u].y, qb = d[u].z;
      for (int idx = lt + 256; idx < nrows * 3; idx += 256) {  // tiles of very low-degree rows (> 85 rows)
        const int a2 = idx / nrows, row2 = r0 + (idx - a2 * nrows);
        const int lo2 = A.inc_ptr[row2] - qb, hi2 = A.inc_ptr[row2 + 1] - qb;
        double s = 0.0;
        for (int j = lo2; j < hi2; ++j) s += sc[a2][j];
        const double* pp = pvec + PS * (int64_t)(A.lo + row2);
        const int i1 = (a2 == 0) ? 1 : (a2 == 1 ? 3 : 4), i2 = (a2 == 2) ? 5 : (a2 == 1 ? 4 : 2);
        double dg = A.hd[(int64_t)a2 * n + row2] * pp[0];
        dg += A.hd[(int64_t)i1 * n + row2] * pp[1];
        dg += A.hd[(int64_t)i2 * n + row2] * pp[2];
        if (WITH_D2) dg += A.d2[3 * (int64_t)row2 + a2] * pp[a2];
        s += dg;
        yout[3 * (int64_t)row2 + a2] = s;
        dot += pp[a2] * s;
      }
    }
    __syncthreads();
  }
  return solo_sum(dot, red);
}

// z = M^-1 r for the problem's rows with the vector update fused in front:
//   INIT:  r = b, y = 0, z = M^-1 r, p = z          returns (r.z, r.r)
//   else:  y += alpha p, r -= alpha Ap, z = M^-1 r  returns (r.z, r.r)
// Chain preconditioner: one wavefront per 256-row tile (chain_apply_lean<4>), the 8 waves take the tiles round-robin.
template <bool INIT>
__device__ __forceinline__ void solo_precond(const SoloArgs& S, const SoloProb& P, double alpha, double* tile_lds,
                                             double* red, double& rz_out, double& rr_out) {
  constexpr int CH = SOLO_CH, TILE = 64 * CH, STRIDE = 3 * CH + 1, NV = 3 * CH;
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const CgVec& V = S.V;
  double rz = 0.0, rr = 0.0;
  const int64_t np = S.C.n_pad;
  double* __restrict__ vy = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vz = V.z;
  const double* __restrict__ vap = V.ap;
  double* __restrict__ pown = V.p + PS * (int64_t)V.lo;
  if (S.C.cw != nullptr) {
    double* buf = tile_lds + wave * (64 * STRIDE);
    double* ch = buf + lane * STRIDE;
    const int n_ct = (P.nrows + TILE - 1) / TILE;
    for (int t = wave; t < n_ct; t += SOLO_WG / 64) {
      const int64_t wbase = (int64_t)P.row0 + (int64_t)t * TILE, f0 = 3 * wbase;
      const int rows_here = min(TILE, P.nrows - t * TILE);
      const unsigned lim = 3u * (unsigned)rows_here;
      const double* cw_tile = S.C.cw + wbase;
      double W[CH][9];
#pragma unroll
      for (int k = 0; k < CH; ++k)
#pragma unroll
        for (int c = 0; c < 9; ++c) W[k][c] = (cw_tile + ((int64_t)c * np + k * 64))[lane];
      double rv[NV];
      if (INIT) {
        const double* bt = S.b + f0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const unsigned e = lane + 64u * j;
          const double v = bt[e < lim ? e : 0u];   // clamped address: no load behind a branch
          rv[j] = e < lim ? v : 0.0;
        }
      } else {
        double av[NV], yv[NV], pv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const unsigned e = lane + 64u * j, ec = e < lim ? e : 0u;   // clamped addresses: no load behind a branch
          const bool ok = e < lim;
          const double r_ = vr[f0 + ec], a_ = vap[f0 + ec], y_ = vy[f0 + ec], p_ = pown[f0 + ec];
          rv[j] = ok ? r_ : 0.0;
          av[j] = ok ? a_ : 0.0;
          yv[j] = ok ? y_ : 0.0;
          pv[j] = ok ? p_ : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const unsigned e = lane + 64u * j;
          rv[j] -= alpha * av[j];
          yv[j] += alpha * pv[j];
          if (e < lim) {
            vr[f0 + e] = rv[j];
            vy[f0 + e] = yv[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        rr += rv[j] * rv[j];
        buf[e + e / (3 * CH)] = rv[j];
      }
      wave_lds_sync();
      chain_apply_lean<CH, false>(W, S.C.cs + wbase, np, lane, ch, S.chain_steps, S.scan_levels);
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        const double z = buf[e + e / (3 * CH)];
        if (e < lim) {
          vz[f0 + e] = z;
          if (INIT) {
            vy[f0 + e] = 0.0;
            vr[f0 + e] = rv[j];
            pown[f0 + e] = z;
          }
        }
        rz += rv[j] * z;
      }
      wave_lds_sync();
    }
  } else {  // 3x3 block-Jacobi: thread per row
    const int64_t n = V.n_loc;
    for (int i = tid; i < P.nrows; i += SOLO_WG) {
      const int row = P.row0 + i;
      const int64_t f = 3 * (int64_t)row;
      double r0, r1, r2;
      if (INIT) {
        r0 = S.b[f]; r1 = S.b[f + 1]; r2 = S.b[f + 2];
      } else {
        vy[f] += alpha * pown[f];
        vy[f + 1] += alpha * pown[f + 1];
        vy[f + 2] += alpha * pown[f + 2];
        r0 = vr[f] - alpha * vap[f];
        r1 = vr[f + 1] - alpha * vap[f + 1];
        r2 = vr[f + 2] - alpha * vap[f + 2];
      }
      double z0, z1, z2;
      minv_apply(V.minv, n, row, r0, r1, r2, z0, z1, z2);
      vr[f] = r0; vr[f + 1] = r1; vr[f + 2] = r2;
      vz[f] = z0; vz[f + 1] = z1; vz[f + 2] = z2;
      if (INIT) {
        vy[f] = 0.0; vy[f + 1] = 0.0; vy[f + 2] = 0.0;
        pown[f] = z0; pown[f + 1] = z1; pown[f + 2] = z2;
      }
      rz += r0 * z0 + r1 * z1 + r2 * z2;
      rr += r0 * r0 + r1 * r1 + r2 * r2;
    }
  }
  rz_out = solo_sum(rz, red);
  rr_out = solo_sum(rr, red);
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(SOLO_WG) void k_pcg_solo(SoloArgs S) {
  __shared__ double scr[SOLO_TILES][3][256];                              // 48 KiB: staged block products
  __shared__ double tile_lds[(SOLO_WG / 64) * 64 * (3 * SOLO_CH + 1)];    // 52 KiB: wave-private chain tiles
  __shared__ double red[16];
  const SoloProb P = S.prob[blockIdx.x];
  if (!P.active) return;
  const int tid = threadIdx.x;
  SoloOut* out = S.out + blockIdx.x;
  double* __restrict__ pown = S.V.p + PS * (int64_t)S.V.lo;
  const int64_t f_lo = 3 * (int64_t)P.row0, f_n = 3 * (int64_t)P.nrows;

  double rz, bb, rr;
  solo_precond<true>(S, P, 0.0, tile_lds, red, rz, bb);
  rr = bb;
  const double tol2 = P.rtol * P.rtol * bb;
  int it = 0, done = (bb == 0.0) ? 1 : 0;
  __syncthreads();  // p (global) is read by other threads of this workgroup next
  while (!done && it < P.max_it) {
    const double pap = solo_spmv<true>(S.A, P, S.V.p, S.V.ap, scr, red);
    const double alpha = rz / pap;
    double rz_new;
    solo_precond<false>(S, P, alpha, tile_lds, red, rz_new, rr);
    ++it;
    if (rr <= tol2) {  // converged: leave p alone (k_cg_update2 does the same)
      done = 1;
      rz = rz_new;
      break;
    }
    const double beta = rz_new / rz;
    rz = rz_new;
    for (int64_t i = tid; i < f_n; i += SOLO_WG) pown[f_lo + i] = S.V.z[f_lo + i] + beta * pown[f_lo + i];
    __syncthreads();
  }
  double ydotg = 0.0, yHy = 0.0, step2 = 0.0;
  if (S.cand != nullptr) {
    // model decrease and candidate (TrustRegionMinimizer): y.g - y.(H y)/2 needs H y without the LM diagonal
    __syncthreads();
    for (int64_t i = tid; i < f_n; i += SOLO_WG) pown[f_lo + i] = S.V.y[f_lo + i];   // the gather vector now holds y
    __syncthreads();
    yHy = solo_spmv<false>(S.A, P, S.V.p, S.V.ap, scr, red);
    const int64_t off = 3 * (int64_t)S.V.lo + f_lo;
    for (int64_t i = tid; i < f_n; i += SOLO_WG) {
      const double yi = S.V.y[f_lo + i];
      ydotg += yi * S.b[f_lo + i];
      const double dlt = -S.scale[off + i] * yi;
      S.cand[off + i] = S.x[off + i] + dlt;
      step2 += dlt * dlt;
    }
    ydotg = solo_sum(ydotg, red);
    step2 = solo_sum(step2, red);
  }
  if (tid == 0) {
    out->rz = rz;
    out->bb = bb;
    out->rr = rr;
    out->ydotg = ydotg;
    out->yHy = yHy;
    out->step2 = step2;
    out->iters = it;
    out->done = done;
  }
}

}  // namespace dev
}  // namespace pgo
