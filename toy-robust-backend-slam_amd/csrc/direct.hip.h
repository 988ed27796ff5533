// Direct solve of the LM system (H + D'D) y = g for chain-like graphs (the reference's own datasets: INTEL, MIT, CSAIL,
// FR079 -- BASELINE configs[0]-[2]; any long odometry chain with up to ~2000 other edges), standing in for the reference's
// SPARSE_NORMAL_CHOLESKY (main.cpp:154-163) where block-Jacobi PCG needs hundreds of latency-bound iterations per LM
// iteration.
//
// With S the Jacobi column scaling, split the edges into the odometry CHAIN (one edge per consecutive pose pair) and
// the rest (loop closures, bogus loops, extra short-range edges: m of them):
//     H + D'D  =  T + V'V,     T = sum over chain edges (J_e S)'(J_e S) + D'D   (block tridiagonal, SPD, anchored by the
//                                                                               constant pose),
//                              V = the 3m x 3N matrix of the scaled Jacobian rows of the other edges.
// Woodbury:  y = t - Z w,  t = T^-1 g,  Z = T^-1 V',  (I + V Z) w = V t.   Steps, all in fp64 and in a fixed order:
//     k_dlr_setup      per pose the blocks M_i, C_i of T; per low-rank edge its two scaled 3x3 Jacobian blocks
//     k_dlr_factor     block LDL' of T:  W_i = C_i S_{i-1}^-1,  S_i = M_i - W_i C_i'  -- one wavefront per PIECE of the chain
//                      between separator poses (nested dissection; the separators come back through k_dlr_sep_*)
//     k_dlr_prefix, k_dlr_fwd / _mid / _fix
//                      Z and t: one lane per right-hand side, the sweeps over the chain cut into 32 segments side by side
//     k_dlr_sep_*      the separators' Schur complement (order 3 per separator, SPD) and its correction of every column
//     k_dlr_cap        capacitance matrix I + V Z (dense, order 3m) and the right-hand side V t
//     k_chol_panel     left-looking blocked Cholesky of it with the explicit inverse (32 x 32 blocks, one launch per block
//                      column, the block products on the fp64 matrix cores)
//     k_tri_apply      the two triangular solves as products with the inverse factor
//     k_dlr_combine    y = t - Z w
// followed by a step of iterative refinement against the assembled block-CSR matrix (k_spmv), which brings the solution to
// the accuracy of a backward-stable direct solve (the capacitance matrix reaches condition 1e7 at large trust-region
// radii).  Restated in numpy and checked against the oracle's SuperLU solve: tests/test_direct_solve_math.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgo {
namespace dev {

constexpr int DLR_REC = 16;  // doubles per pose: input record M (6: 00 01 02 11 12 22) | C (9, row-major) | pad;
                             // factor record W (9, row-major) | S^-1 (6) | pad
constexpr int DLR_V = 18;    // doubles per low-rank edge: Xa (9) | Xb (9), X[k][c] = d e_k / d (pose)_c * scale_c
constexpr int CHOL_NB = 32;
constexpr int DLR_MAX_SEP = 15;  // separator poses: the chain factorisation runs nsep + 1 pieces side by side (3 separators up to ~2000 poses)

struct DlrArgs {
  int32_t n;                  // poses
  int32_t m;                  // low-rank edges
  int32_t K;                  // 3 m
  int32_t Kp;                 // K rounded up to a multiple of CHOL_NB (order of the padded capacitance matrix)
  int32_t ld;                 // row stride of T / Z (>= K + 1, a multiple of 64)
  const double* jr;           // edge records (REC doubles)
  const double* scale;        // [n x 3]
  const double* d2;           // [n x 3] LM diagonal
  const int32_t* e_ia;
  const int32_t* e_ib;
  const int32_t* chain_edge;  // [n]: local edge joining (i, i + 1), -1 for the last pose
  const int32_t* lr_edge;     // [m]
  const int32_t* va;          // [m] endpoints of the low-rank edges
  const int32_t* vb;
  double* trec;               // [n][DLR_REC]
  double* fac;                // [n][DLR_REC]
  double* vrec;               // [m][DLR_V]
  double* Z;                  // [3n][ld]
  double* cap;                // [Kp][Kp]
  double* dwork;              // [Kp / 32][32][32]: the diagonal blocks of the capacitance matrix, updated by k_chol_panel
  double* cvec;               // [Kp]
  // separators of the chain (k_dlr_factor runs the pieces between them side by side, see k_dlr_sep_*)
  int32_t nsep;
  int32_t sep[DLR_MAX_SEP];
  double* ksep;               // [nsep][18]: the couplings taken out of T: C_s = T[s][s-1] (9) | C_{s+1} = T[s+1][s] (9)
  // METHOD 2 (switches eliminated per edge, k_assemble<true>): d e / d s per edge and the elimination coefficient c; null otherwise
  const double* sw_js;
  const double* sw_c;
  int32_t rec_n;              // doubles per edge record (REC, or REC_INFO in the information-weighted mode)
  int32_t rec_info;           // 1: information-weighted records (the last column of d e / d P2 is stored: b3)
};

// scaled Jacobian blocks of edge e (record layouts: kernels.hip.h REC / REC_INFO; the second block is implied up to its
// last column).  METHOD 2: the edge's switch has been eliminated, its term of the pose system is X'(I - c j j')X
// (k_assemble<true>); with beta = (sqrt(1 - c |j|^2) - 1) / |j|^2 the factor (I + beta j j') X has exactly that Gram
// matrix (c |j|^2 < 1 by construction), so the edge stays ONE positive semi-definite low-rank term.
__device__ __forceinline__ void dlr_blocks(const DlrArgs& A, int64_t e, const double* __restrict__ sa, const double* __restrict__ sb,
                                           double (&Xa)[9], double (&Xb)[9]) {
  const double* R = A.jr + e * A.rec_n;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a0 = R[3 * k], a1 = R[3 * k + 1], a2 = R[3 * k + 2];
    const double b2 = A.rec_info ? R[9 + k] : ((k == 2) ? R[9] : 0.0);
    Xa[3 * k] = a0 * sa[0];
    Xa[3 * k + 1] = a1 * sa[1];
    Xa[3 * k + 2] = a2 * sa[2];
    Xb[3 * k] = -a0 * sb[0];
    Xb[3 * k + 1] = -a1 * sb[1];
    Xb[3 * k + 2] = b2 * sb[2];
  }
  if (A.sw_c != nullptr) {
    const double cc = A.sw_c[e];
    if (cc != 0.0) {
      const double j0 = A.sw_js[3 * e], j1 = A.sw_js[3 * e + 1], j2 = A.sw_js[3 * e + 2];
      const double jj = j0 * j0 + j1 * j1 + j2 * j2;
      const double beta = jj > 0.0 ? (sqrt(fmax(0.0, 1.0 - cc * jj)) - 1.0) / jj : 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double da = beta * (j0 * Xa[c] + j1 * Xa[3 + c] + j2 * Xa[6 + c]);
        const double db = beta * (j0 * Xb[c] + j1 * Xb[3 + c] + j2 * Xb[6 + c]);
        Xa[c] += da * j0; Xa[3 + c] += da * j1; Xa[6 + c] += da * j2;
        Xb[c] += db * j0; Xb[3 + c] += db * j1; Xb[6 + c] += db * j2;
      }
    }
  }
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_dlr_setup(DlrArgs A) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < A.n) {
    const int i = idx;
    double M[6] = {A.d2[3 * (int64_t)i], 0.0, 0.0, A.d2[3 * (int64_t)i + 1], 0.0, A.d2[3 * (int64_t)i + 2]};
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int which = 0; which < 2; ++which) {  // 0: the edge (i - 1, i), 1: the edge (i, i + 1)
      const int e = which == 0 ? (i > 0 ? A.chain_edge[i - 1] : -1) : A.chain_edge[i];
      if (e < 0) continue;
      const int a = A.e_ia[e], b = A.e_ib[e];
      double Xa[9], Xb[9];
      dlr_blocks(A, e, A.scale + 3 * (int64_t)a, A.scale + 3 * (int64_t)b, Xa, Xb);
      const bool self_is_a = (a == i);
      double Xi[9], Xo[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        Xi[k] = self_is_a ? Xa[k] : Xb[k];
        Xo[k] = self_is_a ? Xb[k] : Xa[k];
      }
      M[0] += Xi[0] * Xi[0] + Xi[3] * Xi[3] + Xi[6] * Xi[6];
      M[1] += Xi[0] * Xi[1] + Xi[3] * Xi[4] + Xi[6] * Xi[7];
      M[2] += Xi[0] * Xi[2] + Xi[3] * Xi[5] + Xi[6] * Xi[8];
      M[3] += Xi[1] * Xi[1] + Xi[4] * Xi[4] + Xi[7] * Xi[7];
      M[4] += Xi[1] * Xi[2] + Xi[4] * Xi[5] + Xi[7] * Xi[8];
      M[5] += Xi[2] * Xi[2] + Xi[5] * Xi[5] + Xi[8] * Xi[8];
      if (which == 0) {  // C_i = H_{i,i-1} = (J_i)'(J_{i-1})
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) C[3 * r + c] = Xi[r] * Xo[c] + Xi[3 + r] * Xo[3 + c] + Xi[6 + r] * Xo[6 + c];
      }
    }
    // separator s: the couplings (s, s-1) and (s+1, s) leave T (they come back through k_dlr_sep_*), so that the chain
    // falls into independently factorised pieces; the diagonal blocks keep every edge's contribution (anchored pieces)
    int cut = -1;
    for (int j = 0; j < A.nsep; ++j) {
      if (i == A.sep[j]) cut = 2 * j;
      if (i == A.sep[j] + 1) cut = 2 * j + 1;
    }
    double* o = A.trec + (int64_t)i * DLR_REC;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = M[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[6 + k] = cut >= 0 ? 0.0 : C[k];
    o[15] = 0.0;
    if (cut >= 0) {
#pragma unroll
      for (int k = 0; k < 9; ++k) A.ksep[(int64_t)cut * 9 + k] = C[k];
    }
  } else if (idx < A.n + A.m) {
    const int j = idx - A.n;
    const int e = A.lr_edge[j];
    const int a = A.e_ia[e], b = A.e_ib[e];
    double Xa[9], Xb[9];
    dlr_blocks(A, e, A.scale + 3 * (int64_t)a, A.scale + 3 * (int64_t)b, Xa, Xb);
    double* o = A.vrec + (int64_t)j * DLR_V;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      o[k] = Xa[k];
      o[9 + k] = Xb[k];
    }
  }
}

// Block LDL' of the block-tridiagonal T, one wavefront (every lane runs the same recurrence; lane 0 stores).  The input
// records pass through LDS in chunks of 128 poses, the next chunk is fetched while the current one is factorised: the
// 3x3 recurrence itself (a cofactor inverse and two 3x3 products per pose, ~0.1 us) is the critical path.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(64) void k_dlr_factor(const double* __restrict__ trec_all, int n_all, double* __restrict__ fac_all, DlrArgs A) {
  constexpr int CH = 128;
  __shared__ double buf[2][CH * DLR_REC];
  const int lane = threadIdx.x;
  // piece b = [start, end): pieces begin at a separator (C = 0 there AND at the next pose, so the recurrence restarts twice)
  const int start = blockIdx.x == 0 ? 0 : A.sep[blockIdx.x - 1];
  const int end = (int)blockIdx.x < A.nsep ? A.sep[blockIdx.x] : n_all;
  const int n = end - start;
  const double* __restrict__ trec = trec_all + (int64_t)start * DLR_REC;
  double* __restrict__ fac = fac_all + (int64_t)start * DLR_REC;
  const int n_chunks = (n + CH - 1) / CH;
  const int64_t total = (int64_t)n * DLR_REC;
  for (int k = 0; k < CH * DLR_REC / 64; ++k) {
    const int64_t idx = (int64_t)k * 64 + lane;
    buf[0][k * 64 + lane] = idx < total ? trec[idx] : 0.0;
  }
  __syncthreads();
  double si0 = 0, si1 = 0, si2 = 0, si3 = 0, si4 = 0, si5 = 0;  // S_{i-1}^-1 (00 01 02 11 12 22)
  for (int c = 0; c < n_chunks; ++c) {
    double pre[CH * DLR_REC / 64];
    if (c + 1 < n_chunks) {
#pragma unroll
      for (int k = 0; k < CH * DLR_REC / 64; ++k) {
        const int64_t idx = (int64_t)(c + 1) * CH * DLR_REC + (int64_t)k * 64 + lane;
        pre[k] = idx < total ? trec[idx] : 0.0;
      }
    }
    const double* B = buf[c & 1];
    const int cnt = min(CH, n - c * CH);
    for (int ii = 0; ii < cnt; ++ii) {
      const double* R = B + ii * DLR_REC;
      const double m0 = R[0], m1 = R[1], m2 = R[2], m3 = R[3], m4 = R[4], m5 = R[5];
      const double c0 = R[6], c1 = R[7], c2 = R[8], c3 = R[9], c4 = R[10], c5 = R[11], c6 = R[12], c7 = R[13], c8 = R[14];
      // W = C S_{i-1}^-1   (S^-1 symmetric); zero for the first pose (si = 0, C = 0)
      const double w0 = c0 * si0 + c1 * si1 + c2 * si2, w1 = c0 * si1 + c1 * si3 + c2 * si4, w2 = c0 * si2 + c1 * si4 + c2 * si5;
      const double w3 = c3 * si0 + c4 * si1 + c5 * si2, w4 = c3 * si1 + c4 * si3 + c5 * si4, w5 = c3 * si2 + c4 * si4 + c5 * si5;
      const double w6 = c6 * si0 + c7 * si1 + c8 * si2, w7 = c6 * si1 + c7 * si3 + c8 * si4, w8 = c6 * si2 + c7 * si4 + c8 * si5;
      // S = M - W C'
      const double a00 = m0 - (w0 * c0 + w1 * c1 + w2 * c2), a01 = m1 - (w0 * c3 + w1 * c4 + w2 * c5);
      const double a02 = m2 - (w0 * c6 + w1 * c7 + w2 * c8), a11 = m3 - (w3 * c3 + w4 * c4 + w5 * c5);
      const double a12 = m4 - (w3 * c6 + w4 * c7 + w5 * c8), a22 = m5 - (w6 * c6 + w7 * c7 + w8 * c8);
      const double k00 = a11 * a22 - a12 * a12, k01 = a12 * a02 - a01 * a22, k02 = a01 * a12 - a11 * a02;
      const double id = 1.0 / (a00 * k00 + a01 * k01 + a02 * k02);
      si0 = k00 * id;
      si1 = k01 * id;
      si2 = k02 * id;
      si3 = (a00 * a22 - a02 * a02) * id;
      si4 = (a01 * a02 - a00 * a12) * id;
      si5 = (a00 * a11 - a01 * a01) * id;
      if (lane == 0) {
        double2* o = reinterpret_cast<double2*>(fac + (int64_t)(c * CH + ii) * DLR_REC);
        o[0] = make_double2(w0, w1);
        o[1] = make_double2(w2, w3);
        o[2] = make_double2(w4, w5);
        o[3] = make_double2(w6, w7);
        o[4] = make_double2(w8, si0);
        o[5] = make_double2(si1, si2);
        o[6] = make_double2(si3, si4);
        o[7] = make_double2(si5, 0.0);
      }
    }
    if (c + 1 < n_chunks) {
      double* Bn = buf[(c + 1) & 1];
#pragma unroll
      for (int k = 0; k < CH * DLR_REC / 64; ++k) Bn[k * 64 + lane] = pre[k];
    }
    __syncthreads();
  }
}

// T x = rhs for many right-hand sides at once.  A column's two sweeps over the chain are 2n dependent steps; they are cut
// into `nseg` segments of `seglen` poses that run side by side: every (segment, column) pair sweeps its segment from a
// zero state, and because the recurrences are affine in the incoming state,
//     t_i = t_i(local) + G_i t_in,     G_i  = (-W_i) G_{i-1}         (G  = I before the segment's first pose),
//     x_i = x_i(local) + Gb_i x_in,    Gb_i = (-W_{i+1}') Gb_{i+1}   (Gb = I after the segment's last pose),
// the true values follow from the segment-end states by a short serial recursion over the segments (<= 31 steps of one
// 3x3 product) and one independent update per pose.  G / Gb depend on the factorisation only (k_dlr_prefix).
//     k_dlr_fwd   local forward sweep                       -> X (t local), E  (t at the segment ends)
//     k_dlr_mid   incoming t, local backward sweep          -> X (x local), E2 (x at the segment starts)
//     k_dlr_fix   incoming x, x_i += Gb_i x_in
// Column c < K is row c of V (non-zero on the two endpoint poses of its edge); column `vec_col` is the dense vector
// rhs_b - rhs_sub (rhs_sub may be null).  One lane per column, a workgroup = 256 columns of one segment.
constexpr int DLR_PRE = 18;   // doubles per pose of the prefix products: G (9) | Gb (9), row-major
constexpr int DLR_MAX_SEG = 64;
struct DlrColsArgs {
  const double* fac;
  const double* pre;
  int32_t n, ncols, K, vec_col, ld, nseg, seglen;
  int32_t ucol0, nsep;         // columns ucol0 .. ucol0 + 3 nsep - 1: the couplings of the separator poses (k_dlr_sep_*)
  int32_t sep[DLR_MAX_SEP];
  const double* ksep;
  const double* vrec;
  const int32_t* va;
  const int32_t* vb;
  const double* rhs_b;
  const double* rhs_sub;
  double* X;   // [3n][ld]
  double* E;   // [nseg][3][ld]
  double* E2;  // [nseg][3][ld]
};

// prefix products of one segment per thread: the first half of the workgroup forward (G), the second half backward (Gb)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(512) void k_dlr_prefix(const double* __restrict__ fac, int n, int nseg, int seglen, double* __restrict__ pre) {
  const int t = threadIdx.x;
  const int half = blockDim.x >> 1;
  const bool back = t >= half;
  const int s = back ? t - half : t;
  if (s >= nseg) return;
  const int i0 = s * seglen, i1 = min(n, i0 + seglen);
  double g[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (!back) {
#pragma unroll 8
    for (int i = i0; i < i1; ++i) {
      const double* W = fac + (int64_t)i * DLR_REC;
      double h[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[3 * r + c] = -(W[3 * r] * g[c] + W[3 * r + 1] * g[3 + c] + W[3 * r + 2] * g[6 + c]);
      double* o = pre + (int64_t)i * DLR_PRE;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        g[k] = h[k];
        o[k] = h[k];
      }
    }
  } else {
#pragma unroll 8
    for (int i = i1 - 1; i >= i0; --i) {
      const double* W = fac + (int64_t)(i + 1) * DLR_REC;  // record n is all zero
      double h[9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[3 * r + c] = -(W[r] * g[c] + W[3 + r] * g[3 + c] + W[6 + r] * g[6 + c]);
      double* o = pre + (int64_t)i * DLR_PRE + 9;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        g[k] = h[k];
        o[k] = h[k];
      }
    }
  }
}

struct DlrCol {   // what a lane knows about its column
  int a, b;
  double ra0, ra1, ra2, rb0, rb1, rb2;
  bool act, isvec;
};
__device__ __forceinline__ DlrCol dlr_col(const DlrColsArgs& A, int col) {
  DlrCol c;
  c.act = col < A.ncols;
  c.isvec = c.act && col == A.vec_col;
  c.a = c.b = -1;
  c.ra0 = c.ra1 = c.ra2 = c.rb0 = c.rb1 = c.rb2 = 0.0;
  if (c.act && !c.isvec && col < A.K) {
    const int j = col / 3, k = col - 3 * j;
    c.a = A.va[j];
    c.b = A.vb[j];
    const double* v = A.vrec + (int64_t)j * DLR_V + 3 * k;
    c.ra0 = v[0]; c.ra1 = v[1]; c.ra2 = v[2];
    c.rb0 = v[9]; c.rb1 = v[10]; c.rb2 = v[11];
  } else if (c.act && !c.isvec && col >= A.ucol0 && col < A.ucol0 + 3 * A.nsep) {
    // column (j, k) of B = the part of T's column 3 s_j + k that lies in the pieces: C_s'[:, k] at pose s - 1, C_{s+1}[:, k] at s + 1
    const int u = col - A.ucol0, j = u / 3, k = u - 3 * j;
    const double* ks = A.ksep + (int64_t)j * 18;
    c.a = A.sep[j] - 1;
    c.b = A.sep[j] + 1;
    c.ra0 = ks[3 * k]; c.ra1 = ks[3 * k + 1]; c.ra2 = ks[3 * k + 2];
    c.rb0 = ks[9 + k]; c.rb1 = ks[12 + k]; c.rb2 = ks[15 + k];
  }
  return c;
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_fwd(DlrColsArgs A) {
  constexpr int CH = 64;
  __shared__ double fb[CH * DLR_REC];
  __shared__ double rv[CH * 3];   // the dense right-hand side of the chunk (read by the one lane that owns that column: from
                                  // global memory its loads sat between that lane's stores, one round trip per pose)
  const int tid = threadIdx.x;
  const int col = blockIdx.x * 256 + tid;
  const int s = blockIdx.y;
  const int i0 = s * A.seglen, i1 = min(A.n, i0 + A.seglen);
  const DlrCol c = dlr_col(A, col);
  const bool vec_here = A.vec_col >= (int)blockIdx.x * 256 && A.vec_col < (int)blockIdx.x * 256 + 256 && A.vec_col < A.ncols;
  const int64_t ld = A.ld;
  double* __restrict__ X = A.X + col;
  double t0 = 0, t1 = 0, t2 = 0;
  for (int c0 = i0; c0 < i1; c0 += CH) {
    const int cnt = min(CH, i1 - c0);
    __syncthreads();
    for (int idx = tid; idx < cnt * DLR_REC; idx += 256) fb[idx] = A.fac[(int64_t)c0 * DLR_REC + idx];
    if (vec_here && tid < cnt * 3) {
      double v = A.rhs_b[3 * (int64_t)c0 + tid];
      if (A.rhs_sub) v -= A.rhs_sub[3 * (int64_t)c0 + tid];
      rv[tid] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int ii = 0; ii < cnt; ++ii) {
      const int i = c0 + ii;
      double r0 = 0, r1 = 0, r2 = 0;
      if (i == c.a) { r0 = c.ra0; r1 = c.ra1; r2 = c.ra2; }
      if (i == c.b) { r0 += c.rb0; r1 += c.rb1; r2 += c.rb2; }
      if (c.isvec) { r0 = rv[3 * ii]; r1 = rv[3 * ii + 1]; r2 = rv[3 * ii + 2]; }
      const double* W = fb + ii * DLR_REC;
      const double n0 = r0 - (W[0] * t0 + W[1] * t1 + W[2] * t2);
      const double n1 = r1 - (W[3] * t0 + W[4] * t1 + W[5] * t2);
      const double n2 = r2 - (W[6] * t0 + W[7] * t1 + W[8] * t2);
      t0 = n0; t1 = n1; t2 = n2;
      if (c.act) {
        X[(3 * (int64_t)i) * ld] = t0;
        X[(3 * (int64_t)i + 1) * ld] = t1;
        X[(3 * (int64_t)i + 2) * ld] = t2;
      }
    }
  }
  if (c.act) {
    double* e = A.E + (int64_t)s * 3 * ld + col;
    e[0] = t0; e[ld] = t1; e[2 * ld] = t2;
  }
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_mid(DlrColsArgs A) {
  constexpr int CH = 64;
  __shared__ double fb[(CH + 1) * DLR_REC];
  __shared__ double gb[CH * 9];
  const int tid = threadIdx.x;
  const int col = blockIdx.x * 256 + tid;
  const int s = blockIdx.y;
  const int n = A.n;
  const int i0 = s * A.seglen, i1 = min(n, i0 + A.seglen);
  const bool act = col < A.ncols;
  const int64_t ld = A.ld;
  double* __restrict__ X = A.X + col;
  // t entering this segment: t_in(q + 1) = E(q) + F_q t_in(q),  F_q = G at the last pose of segment q
  double ti0 = 0, ti1 = 0, ti2 = 0;
  for (int q0 = 0; q0 < s; q0 += 16) {   // 16 segments' end states requested together (the chain over q is 3 FMAs per step)
    double ev[16][3];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double* e = A.E + (int64_t)min(q0 + u, s - 1) * 3 * ld + col;
#pragma unroll
      for (int c = 0; c < 3; ++c) ev[u][c] = act ? e[c * ld] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int q = q0 + u;
      if (q < s) {
        const double* F = A.pre + (int64_t)(min(n, (q + 1) * A.seglen) - 1) * DLR_PRE;
        const double n0 = ev[u][0] + F[0] * ti0 + F[1] * ti1 + F[2] * ti2;
        const double n1 = ev[u][1] + F[3] * ti0 + F[4] * ti1 + F[5] * ti2;
        const double n2 = ev[u][2] + F[6] * ti0 + F[7] * ti1 + F[8] * ti2;
        ti0 = n0; ti1 = n1; ti2 = n2;
      }
    }
  }
  double z0 = 0, z1 = 0, z2 = 0;
  const int n_chunks = (i1 - i0 + CH - 1) / CH;
  for (int cc = n_chunks - 1; cc >= 0; --cc) {
    const int c0 = i0 + cc * CH;
    const int cnt = min(CH, i1 - c0);
    __syncthreads();
    for (int idx = tid; idx < (cnt + 1) * DLR_REC; idx += 256) fb[idx] = A.fac[(int64_t)c0 * DLR_REC + idx];  // fac has n + 1 records
    for (int idx = tid; idx < cnt * 9; idx += 256) {
      const int ii = idx / 9;
      gb[idx] = A.pre[(int64_t)(c0 + ii) * DLR_PRE + (idx - 9 * ii)];
    }
    __syncthreads();
    for (int b1 = cnt; b1 > 0; b1 -= 16) {   // poses b1-1 .. b1-16 of the chunk, their loads issued together
      double tl[16][3];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = c0 + max(b1 - 1 - u, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) tl[u][c] = act ? X[(3 * (int64_t)i + c) * ld] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ii = b1 - 1 - u;
        if (ii >= 0) {
          const int i = c0 + ii;
          const double* G = gb + ii * 9;
          const double u0 = tl[u][0] + (G[0] * ti0 + G[1] * ti1 + G[2] * ti2);
          const double u1 = tl[u][1] + (G[3] * ti0 + G[4] * ti1 + G[5] * ti2);
          const double u2 = tl[u][2] + (G[6] * ti0 + G[7] * ti1 + G[8] * ti2);
          const double* F = fb + ii * DLR_REC;
          const double* Wn = fb + (ii + 1) * DLR_REC;
          const double n0 = F[9] * u0 + F[10] * u1 + F[11] * u2 - (Wn[0] * z0 + Wn[3] * z1 + Wn[6] * z2);
          const double n1 = F[10] * u0 + F[12] * u1 + F[13] * u2 - (Wn[1] * z0 + Wn[4] * z1 + Wn[7] * z2);
          const double n2 = F[11] * u0 + F[13] * u1 + F[14] * u2 - (Wn[2] * z0 + Wn[5] * z1 + Wn[8] * z2);
          z0 = n0; z1 = n1; z2 = n2;
          if (act) {
            X[(3 * (int64_t)i) * ld] = z0;
            X[(3 * (int64_t)i + 1) * ld] = z1;
            X[(3 * (int64_t)i + 2) * ld] = z2;
          }
        }
      }
    }
  }
  if (act) {
    double* e = A.E2 + (int64_t)s * 3 * ld + col;
    e[0] = z0; e[ld] = z1; e[2 * ld] = z2;
  }
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_fix(DlrColsArgs A) {
  __shared__ double gbs[64 * 9];
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int s = blockIdx.y;
  if (s == A.nseg - 1) return;   // nothing enters the last segment (the whole workgroup leaves)
  const bool live = col < A.ncols;
  const int n = A.n;
  const int i0 = s * A.seglen, i1 = min(n, i0 + A.seglen);
  const int64_t ld = A.ld;
  double* __restrict__ X = A.X + col;
  // x entering this segment from the right: x_in(q - 1) = E2(q) + Fb_q x_in(q),  Fb_q = Gb at the first pose of segment q
  double x0 = 0, x1 = 0, x2 = 0;
  for (int q0 = A.nseg - 1; q0 > s; q0 -= 16) {   // 16 segments' start states requested together
    double ev[16][3];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double* e = A.E2 + (int64_t)max(q0 - u, s + 1) * 3 * ld + col;
#pragma unroll
      for (int c = 0; c < 3; ++c) ev[u][c] = live ? e[c * ld] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int q = q0 - u;
      if (q > s) {
        const double* F = A.pre + (int64_t)(q * A.seglen) * DLR_PRE + 9;
        const double n0 = ev[u][0] + F[0] * x0 + F[1] * x1 + F[2] * x2;
        const double n1 = ev[u][1] + F[3] * x0 + F[4] * x1 + F[5] * x2;
        const double n2 = ev[u][2] + F[6] * x0 + F[7] * x1 + F[8] * x2;
        x0 = n0; x1 = n1; x2 = n2;
      }
    }
  }
  // every pose's update is independent of the others: 16 poses' loads are issued before the first of them is used; Gb comes
  // through LDS (read from global memory its loads sat between the stores: one round trip per pose)
  for (int c0 = i0; c0 < i1; c0 += 64) {
    const int cnt = min(64, i1 - c0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < cnt * 9; idx += 256) {
      const int ii = idx / 9;
      gbs[idx] = A.pre[(int64_t)(c0 + ii) * DLR_PRE + 9 + (idx - 9 * ii)];
    }
    __syncthreads();
    if (!live) continue;
    for (int b0 = 0; b0 < cnt; b0 += 16) {
      double v[16][3];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = c0 + min(b0 + u, cnt - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[u][c] = X[(3 * (int64_t)i + c) * ld];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ii = b0 + u;
        if (ii < cnt) {
          const int i = c0 + ii;
          const double* G = gbs + ii * 9;
          X[(3 * (int64_t)i) * ld] = v[u][0] + (G[0] * x0 + G[1] * x1 + G[2] * x2);
          X[(3 * (int64_t)i + 1) * ld] = v[u][1] + (G[3] * x0 + G[4] * x1 + G[5] * x2);
          X[(3 * (int64_t)i + 2) * ld] = v[u][2] + (G[6] * x0 + G[7] * x1 + G[8] * x2);
        }
      }
    }
  }
}

// ---- the couplings taken out at the separators come back here: nested dissection of the chain.  With the poses ordered
// (pieces, separators),  T = [[Tp, B], [B', Ms]]  where Tp = the pieces (what k_dlr_factor factorised; the batched sweeps
// apply Tt^-1 = blockdiag(Tp^-1, Ms^-1)), B = the couplings C_s, C_{s+1} and Ms = the separators' diagonal blocks:
//     x_s = S^-1 (Ms z_s - B' z_p),   x_p = z_p - Y x_s,      z = Tt^-1 r,   Y = Tp^-1 B,   S = Ms - B' Y
// S (order 3 nsep <= 9) is a Schur complement of the SPD T: SPD, inverted without pivoting.  Y = 3 nsep more columns
// of the batched solve (zero on the separator rows); S^-1 once per factorisation; then every other column j:
//     k_dlr_sep_w      w_j = S^-1 (Ms z_s - C_s z_{s-1} - C_{s+1}' z_{s+1})
//     k_dlr_sep_apply  z_j -= Y w_j on the piece rows,  z_j = w_j on the separator rows
// (a first form restored the couplings as an indefinite low-rank term, T = Tt + U K U', order 27 with pivoting: 72 us for
// the reduced system alone)
constexpr int DLR_MAX_U = 3 * DLR_MAX_SEP;
struct DlrSepArgs {
  int32_t nsep, nU, n;
  int32_t sep[DLR_MAX_SEP];
  const double* ksep;   // [nsep][18]: C_s (9) | C_{s+1} (9), row-major
  const double* trec;   // pose records: Ms = trec[s][0..5]
  const double* Y;      // the B columns inside Z: Y[r * yld + u]
  int32_t yld;
  double* Sinv;         // [nU][nU]
  double* X;            // the columns to correct: [3n][ld]
  int32_t ld, ncols;
  double* Wm;           // [nU][ld]
};

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_sep_system(DlrSepArgs A) {
  __shared__ double Ab[DLR_MAX_U * 2 * DLR_MAX_U];   // [S | I], row stride 2 nU
  const int tid = threadIdx.x, nU = A.nU, w2 = 2 * nU;
  for (int e = tid; e < nU * nU; e += 256) {
    const int u = e / nU, v = e - u * nU;
    const int j = u / 3, c = u - 3 * j, jv = v / 3, cv = v - 3 * jv;
    const int sp = A.sep[j];
    const double* ks = A.ksep + (int64_t)j * 18;
    double val = 0.0;
    if (j == jv) {   // Ms (symmetric: 00 01 02 11 12 22)
      const double* M = A.trec + (int64_t)sp * DLR_REC;
      const int lo = min(c, cv), hi = max(c, cv);
      val = M[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
    }
    // - b_u' Y[:, v] = - sum_k C_s[c][k] Y[3 (s-1) + k][v] - sum_k C_{s+1}[k][c] Y[3 (s+1) + k][v]
    for (int k = 0; k < 3; ++k)
      val -= ks[3 * c + k] * A.Y[(int64_t)(3 * (sp - 1) + k) * A.yld + v] + ks[9 + 3 * k + c] * A.Y[(int64_t)(3 * (sp + 1) + k) * A.yld + v];
    Ab[u * w2 + v] = val;
    Ab[u * w2 + nU + v] = (u == v) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int k = 0; k < nU; ++k) {   // Gauss-Jordan, no pivoting (SPD)
    const double pinv = 1.0 / Ab[k * w2 + k];
    double nv[16];
    int cnt = 0;
    for (int e = tid; e < nU * w2; e += 256, ++cnt) {
      const int r = e / w2, c = e - r * w2;
      const double pk = Ab[k * w2 + c] * pinv;
      nv[cnt] = (r == k) ? pk : Ab[e] - Ab[r * w2 + k] * pk;
    }
    __syncthreads();
    cnt = 0;
    for (int e = tid; e < nU * w2; e += 256, ++cnt) Ab[e] = nv[cnt];
    __syncthreads();
  }
  for (int e = tid; e < nU * nU; e += 256) A.Sinv[e] = Ab[(e / nU) * w2 + nU + (e % nU)];
}

// w_j = S^-1 (Ms z_s - C_s z_{s-1} - C_{s+1}' z_{s+1}) for the columns j < ncols of X
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_sep_w(DlrSepArgs A) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= A.ncols) return;
  const int nU = A.nU;
  double g[DLR_MAX_U];
#pragma unroll
  for (int q = 0; q < DLR_MAX_SEP; ++q) {
    if (q < A.nsep) {
      const int sp = A.sep[q];
      const double* ks = A.ksep + (int64_t)q * 18;
      const double* M = A.trec + (int64_t)sp * DLR_REC;
      double zm[3], zs[3], zp[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        zm[k] = A.X[(int64_t)(3 * (sp - 1) + k) * A.ld + j];
        zs[k] = A.X[(int64_t)(3 * sp + k) * A.ld + j];
        zp[k] = A.X[(int64_t)(3 * (sp + 1) + k) * A.ld + j];
      }
      g[3 * q] = M[0] * zs[0] + M[1] * zs[1] + M[2] * zs[2];
      g[3 * q + 1] = M[1] * zs[0] + M[3] * zs[1] + M[4] * zs[2];
      g[3 * q + 2] = M[2] * zs[0] + M[4] * zs[1] + M[5] * zs[2];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) g[3 * q + c] -= ks[3 * c + k] * zm[k] + ks[9 + 3 * k + c] * zp[k];
    } else {
      g[3 * q] = g[3 * q + 1] = g[3 * q + 2] = 0.0;
    }
  }
#pragma unroll
  for (int u = 0; u < DLR_MAX_U; ++u) {
    if (u < nU) {
      double s = 0.0;
#pragma unroll
      for (int v = 0; v < DLR_MAX_U; ++v) s += (v < nU) ? A.Sinv[u * nU + v] * g[v] : 0.0;
      A.Wm[(int64_t)u * A.ld + j] = s;
    }
  }
}

// z_j -= Y w_j (piece rows; Y is zero on the separator rows), z_j = w_j on the separator rows.  A thread = one column, a
// workgroup = 64 rows of it (w_j in registers, the rows of Y staged in LDS, 8 rows read and written at a time); with a
// single column: one thread per row instead
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_sep_apply(DlrSepArgs A) {
  __shared__ double Ys[64 * DLR_MAX_U];
  __shared__ int sepu[64];   // row -> entry of w that replaces it (-1: a piece row)
  const int nU = A.nU;
  const int r0 = blockIdx.y * 64, r1 = min(3 * A.n, r0 + 64);
  if (threadIdx.x < 64) {
    const int r = r0 + threadIdx.x, pose = r / 3;
    int u = -1;
    for (int q = 0; q < A.nsep; ++q)
      if (pose == A.sep[q]) u = 3 * q + (r - 3 * pose);
    sepu[threadIdx.x] = u;
  }
  for (int e = threadIdx.x; e < 64 * DLR_MAX_U; e += 256) {
    const int rr = e / DLR_MAX_U, b = e - rr * DLR_MAX_U;
    Ys[e] = (r0 + rr < r1 && b < nU) ? A.Y[(int64_t)(r0 + rr) * A.yld + b] : 0.0;
  }
  __syncthreads();
  if (A.ncols == 1) {   // grid.x == 1: threads 0..63 take a row each
    const int r = r0 + threadIdx.x;
    if (threadIdx.x < 64 && r < r1) {
      double s = 0.0;
      for (int b = 0; b < nU; ++b) s += Ys[threadIdx.x * DLR_MAX_U + b] * A.Wm[(int64_t)b * A.ld];
      const int u = sepu[threadIdx.x];
      A.X[(int64_t)r * A.ld] = u >= 0 ? A.Wm[(int64_t)u * A.ld] : A.X[(int64_t)r * A.ld] - s;
    }
    return;
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= A.ncols) return;
  double w[DLR_MAX_U];
#pragma unroll
  for (int b = 0; b < DLR_MAX_U; ++b) w[b] = (b < nU) ? A.Wm[(int64_t)b * A.ld + j] : 0.0;
  for (int b0 = r0; b0 < r1; b0 += 8) {
    double xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = A.X[(int64_t)min(b0 + u, r1 - 1) * A.ld + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = b0 + u;
      if (r < r1) {
        const double* y = Ys + (r - r0) * DLR_MAX_U;
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < DLR_MAX_U; ++b) s += y[b] * w[b];
        const int su = sepu[r - r0];
        double out = xv[u] - s;
        if (su >= 0) {
#pragma unroll
          for (int b = 0; b < DLR_MAX_U; ++b) out = (b == su) ? w[b] : out;
        }
        A.X[(int64_t)r * A.ld + j] = out;
      }
    }
  }
}

// ---- T^-1 r for ONE right-hand side (the refinement's residual) in one launch: the batched kernels above take three
// launches + two for the separators, ~120 us of mostly latency for a single column.  One workgroup, thread s = segment s of
// a FINER segmentation (256 segments, prefix products `pre2`): local forward sweeps, the segment-end states joined by a
// Hillis-Steele scan of the affine maps (a, F) -> (a + F a', F F') in LDS, local backward sweeps on the true t, the
// mirrored scan, the corrections, and the separators' Schur step (k_dlr_sep_*) -- everything between workgroup barriers.
// The column lives in a plain vector x[3n].  Used while a segment has at most 16 poses (n <= 4096).
struct DlrSolve1Args {
  const double* fac;
  const double* pre2;
  int32_t n, nseg, seglen;
  const double* rhs_b;
  const double* rhs_sub;
  double* x;            // [3n]
  int32_t nsep, nU;
  int32_t sep[DLR_MAX_SEP];
  const double* ksep;
  const double* trec;
  const double* Y;
  int32_t yld;
  const double* Sinv;
};
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_solve1(DlrSolve1Args A) {
  __shared__ double sa[2][256 * 3];
  __shared__ double sF[2][256 * 9];
  __shared__ double sg[DLR_MAX_U], sw[DLR_MAX_U];
  const int s = threadIdx.x;
  const int n = A.n, L = A.seglen;
  const int i0 = min(n, s * L), i1 = min(n, i0 + L);
  const bool live = s < A.nseg && i1 > i0;
  // 1. local forward sweep
  double t0 = 0, t1 = 0, t2 = 0;
  for (int i = i0; i < i1; ++i) {
    double r0 = A.rhs_b[3 * i], r1 = A.rhs_b[3 * i + 1], r2 = A.rhs_b[3 * i + 2];
    if (A.rhs_sub) { r0 -= A.rhs_sub[3 * i]; r1 -= A.rhs_sub[3 * i + 1]; r2 -= A.rhs_sub[3 * i + 2]; }
    const double* W = A.fac + (int64_t)i * DLR_REC;
    const double n0 = r0 - (W[0] * t0 + W[1] * t1 + W[2] * t2);
    const double n1 = r1 - (W[3] * t0 + W[4] * t1 + W[5] * t2);
    const double n2 = r2 - (W[6] * t0 + W[7] * t1 + W[8] * t2);
    t0 = n0; t1 = n1; t2 = n2;
    A.x[3 * i] = t0; A.x[3 * i + 1] = t1; A.x[3 * i + 2] = t2;
  }
  // scan of the affine maps: (a, F)_s <- (a_s + F_s a_{s-d}, F_s F_{s-d}); dead segments are the identity map
  auto scan = [&](double (&a)[3], double (&F)[9], bool reverse) {
    int cur = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) sa[0][3 * s + c] = a[c];
#pragma unroll
    for (int c = 0; c < 9; ++c) sF[0][9 * s + c] = F[c];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int o = reverse ? s + d : s - d;
      double na[3], nF[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) na[c] = sa[cur][3 * s + c];
#pragma unroll
      for (int c = 0; c < 9; ++c) nF[c] = sF[cur][9 * s + c];
      if (o >= 0 && o < 256) {
        const double* ao = &sa[cur][3 * o];
        const double* Fo = &sF[cur][9 * o];
        double ta[3], tF[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          ta[r] = na[r] + nF[3 * r] * ao[0] + nF[3 * r + 1] * ao[1] + nF[3 * r + 2] * ao[2];
#pragma unroll
          for (int c = 0; c < 3; ++c) tF[3 * r + c] = nF[3 * r] * Fo[c] + nF[3 * r + 1] * Fo[3 + c] + nF[3 * r + 2] * Fo[6 + c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) na[c] = ta[c];
#pragma unroll
        for (int c = 0; c < 9; ++c) nF[c] = tF[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) sa[cur ^ 1][3 * s + c] = na[c];
#pragma unroll
      for (int c = 0; c < 9; ++c) sF[cur ^ 1][9 * s + c] = nF[c];
      cur ^= 1;
      __syncthreads();
    }
    return cur;
  };
  double a[3] = {t0, t1, t2};
  double F[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (live) {
    const double* G = A.pre2 + (int64_t)(i1 - 1) * DLR_PRE;
#pragma unroll
    for (int c = 0; c < 9; ++c) F[c] = G[c];
  } else {
    a[0] = a[1] = a[2] = 0.0;
  }
  int cur = scan(a, F, false);
  // t entering the segment = the scanned state of the segment before
  double ti0 = 0, ti1 = 0, ti2 = 0;
  if (s > 0) { ti0 = sa[cur][3 * (s - 1)]; ti1 = sa[cur][3 * (s - 1) + 1]; ti2 = sa[cur][3 * (s - 1) + 2]; }
  __syncthreads();
  // 2. local backward sweep on the true t
  double z0 = 0, z1 = 0, z2 = 0;
  for (int i = i1 - 1; i >= i0; --i) {
    const double* G = A.pre2 + (int64_t)i * DLR_PRE;
    const double u0 = A.x[3 * i] + (G[0] * ti0 + G[1] * ti1 + G[2] * ti2);
    const double u1 = A.x[3 * i + 1] + (G[3] * ti0 + G[4] * ti1 + G[5] * ti2);
    const double u2 = A.x[3 * i + 2] + (G[6] * ti0 + G[7] * ti1 + G[8] * ti2);
    const double* Fc = A.fac + (int64_t)i * DLR_REC;
    const double* Wn = A.fac + (int64_t)(i + 1) * DLR_REC;   // record n is all zero
    const double n0 = Fc[9] * u0 + Fc[10] * u1 + Fc[11] * u2 - (Wn[0] * z0 + Wn[3] * z1 + Wn[6] * z2);
    const double n1 = Fc[10] * u0 + Fc[12] * u1 + Fc[13] * u2 - (Wn[1] * z0 + Wn[4] * z1 + Wn[7] * z2);
    const double n2 = Fc[11] * u0 + Fc[13] * u1 + Fc[14] * u2 - (Wn[2] * z0 + Wn[5] * z1 + Wn[8] * z2);
    z0 = n0; z1 = n1; z2 = n2;
    A.x[3 * i] = z0; A.x[3 * i + 1] = z1; A.x[3 * i + 2] = z2;
  }
  a[0] = live ? z0 : 0.0; a[1] = live ? z1 : 0.0; a[2] = live ? z2 : 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) F[c] = (c % 4 == 0) ? 1.0 : 0.0;
  if (live) {
    const double* G = A.pre2 + (int64_t)i0 * DLR_PRE + 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) F[c] = G[c];
  }
  cur = scan(a, F, true);
  double x0 = 0, x1 = 0, x2 = 0;
  if (s + 1 < 256) { x0 = sa[cur][3 * (s + 1)]; x1 = sa[cur][3 * (s + 1) + 1]; x2 = sa[cur][3 * (s + 1) + 2]; }
  // 3. x_i += Gb_i x_in
  for (int i = i0; i < i1; ++i) {
    const double* G = A.pre2 + (int64_t)i * DLR_PRE + 9;
    A.x[3 * i] += G[0] * x0 + G[1] * x1 + G[2] * x2;
    A.x[3 * i + 1] += G[3] * x0 + G[4] * x1 + G[5] * x2;
    A.x[3 * i + 2] += G[6] * x0 + G[7] * x1 + G[8] * x2;
  }
  if (A.nsep == 0) return;
  // 4. the separators: w = S^-1 (Ms x_s - C_s x_{s-1} - C_{s+1}' x_{s+1}),  x -= Y w,  x_s = w
  __syncthreads();   // (the workgroup's global writes above are visible to all its threads after the barrier)
  if (s < A.nU) {
    const int j = s / 3, c = s - 3 * j, sp = A.sep[j];
    const double* ks = A.ksep + (int64_t)j * 18;
    const double* M = A.trec + (int64_t)sp * DLR_REC;
    const double* xm = A.x + 3 * (sp - 1);
    const double* xs = A.x + 3 * sp;
    const double* xp = A.x + 3 * (sp + 1);
    const int m0 = c == 0 ? 0 : (c == 1 ? 1 : 2), m1 = c == 0 ? 1 : (c == 1 ? 3 : 4), m2 = c == 0 ? 2 : (c == 1 ? 4 : 5);
    double g = M[m0] * xs[0] + M[m1] * xs[1] + M[m2] * xs[2];
    for (int k = 0; k < 3; ++k) g -= ks[3 * c + k] * xm[k] + ks[9 + 3 * k + c] * xp[k];
    sg[s] = g;
  }
  __syncthreads();
  if (s < A.nU) {
    double w = 0.0;
    for (int v = 0; v < A.nU; ++v) w += A.Sinv[s * A.nU + v] * sg[v];
    sw[s] = w;
  }
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    int su = -1;
    for (int q = 0; q < A.nsep; ++q)
      if (i == A.sep[q]) su = 3 * q;
    for (int c = 0; c < 3; ++c) {
      const int r = 3 * i + c;
      if (su >= 0) {
        A.x[r] = sw[su + c];
      } else {
        const double* y = A.Y + (int64_t)r * A.yld;
        double acc = 0.0;
        for (int u = 0; u < A.nU; ++u) acc += y[u] * sw[u];
        A.x[r] -= acc;
      }
    }
  }
}

// capacitance matrix I + V Z (row p = row of V, column q = column of Z), its diagonal blocks once more in `dwork`, and
// the right-hand side V t (t = column K of Z); padding rows / columns K .. Kp-1 = identity.
// grid (ld / 256 rounded up, Kp), block 256.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_cap(DlrArgs A) {
  const int p = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q > A.K && q >= A.Kp) return;
  double val = 0.0;
  if (p < A.K && q <= A.K) {
    const int j = p / 3, k = p - 3 * j;
    const int a = A.va[j], b = A.vb[j];
    const double* v = A.vrec + (int64_t)j * DLR_V + 3 * k;
    const double* za = A.Z + (3 * (int64_t)a) * A.ld + q;
    const double* zb = A.Z + (3 * (int64_t)b) * A.ld + q;
    val = v[0] * za[0] + v[1] * za[A.ld] + v[2] * za[2 * (int64_t)A.ld] + v[9] * zb[0] + v[10] * zb[A.ld] + v[11] * zb[2 * (int64_t)A.ld];
  }
  if (q == A.K) A.cvec[p] = val;  // the right-hand side V t (0 on the padding rows)
  if (q < A.Kp) {                 // matrix column q; rows / columns K .. Kp-1 are identity padding
    const double m = ((p < A.K && q < A.K) ? val : 0.0) + (p == q ? 1.0 : 0.0);
    A.cap[(int64_t)p * A.Kp + q] = m;
    if ((p >> 5) == (q >> 5)) A.dwork[(int64_t)(p >> 5) * 1024 + (p & 31) * 32 + (q & 31)] = m;
  }
}

// Left-looking blocked Cholesky C = L L' of the capacitance matrix (32 x 32 blocks) together with the explicit inverse
// N = L^-1 (the two triangular solves then are two dense, fully parallel products instead of 2 x nb dependent steps in
// one workgroup: 200 us -> a few us per solve; the refinement against the assembled matrix absorbs the difference
// between substitution and a product with the inverse).  One launch per block column kb; workgroups of 5 wavefronts:
//   wavefronts 4, 5  factorise the diagonal block kb (its updates are complete: `dwork`) and invert the factor, every
//                    workgroup for itself -- nobody waits for another workgroup -- while
//   wavefronts 0-3   accumulate the workgroup's block: the j are dealt to the four wavefronts, tiles pass through
//                    wave-private LDS (next tiles requested before the current ones are multiplied), the 32 x 32 x 32
//                    products on the matrix cores (v_mfma_f64_16x16x4_f64), the four partial sums added in a fixed order;
//   type 0 workgroups (block row i > kb of L):  P = C_ik - sum_{j<kb} L_ij L_kj',  L_ik = P L_kk^-T,  dwork_i -= L_ik L_ik'
//   type 1 workgroups (block column k < kb of row kb of N):  N_kb,k = -L_kk^-1 sum_{j=k}^{kb-1} L_kb,j N_jk
// Workgroup 0 stores L_kk^-1 = N_kk.
constexpr int CHOL_T = 32 * 33;                                 // doubles of a padded 32 x 32 LDS tile
constexpr int CHOL_LDS_DOUBLES = 2 * CHOL_T + 32 + 8 * CHOL_T + 33 * 32 + 2;  // Dm | Li | dinv | 4 waves x (A | B) tiles | Lt columns + 1 / sqrt(D) | counter
constexpr size_t CHOL_LDS_BYTES = (size_t)CHOL_LDS_DOUBLES * sizeof(double);
constexpr int CHOL_THREADS = 384;

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {   // src_lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_panel(double* __restrict__ Cm, double* __restrict__ Nm, double* __restrict__ dwork,
                                                              int ld, int nb, int kb) {
  extern __shared__ double sm[];
  double* Dm = sm;                 // [32][33]
  double* Li = sm + CHOL_T;        // [32][33]
  double* dinv = sm + 2 * CHOL_T;  // [32]
  double* tiles = sm + 2 * CHOL_T + 32;
  double* colL = tiles + 8 * CHOL_T;                       // [33][32]: the columns of Lt as they are finished, then 1 / sqrt(D)
  int* pflag = reinterpret_cast<int*>(colL + 33 * 32);     // columns handed over so far (33: the scaling too)
  const int tid = threadIdx.x;
  const int w = tid >> 6, lane = tid & 63;
  const int n_chol = nb - 1 - kb;
  const bool type1 = (int)blockIdx.x >= n_chol;
  const int i = kb + 1 + blockIdx.x;          // type 0: block row of L
  const int kcol = (int)blockIdx.x - n_chol;  // type 1: block column of N
  const bool has_work = type1 ? (kcol < kb) : (i < nb);
  // the workgroup's own entries of C_ik and of its diagonal block are requested now, used after the accumulation
  const int pr = (tid & 255) >> 3, pc = 4 * (tid & 7);
  double c_own[4] = {0.0, 0.0, 0.0, 0.0}, d_own[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < 256 && has_work && !type1) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      c_own[v] = Cm[(int64_t)(32 * i + pr) * ld + 32 * kb + pc + v];
      d_own[v] = dwork[(int64_t)i * 1024 + pr * 32 + pc + v];
    }
  }
  if (tid == 0) *pflag = 0;
  __syncthreads();
  if (w == 4) {
    // ---- diagonal block: L_kk (Dm) and its inverse (Li): L D L' and the inverse of the unit triangle, first written as
    // ONE 32-step loop with the rows of both in registers (lane r: row r of the block and row r of N), v_readlane
    // broadcasts only.  Step k scales column k, updates the trailing entries a_rc -= Lt_rk (Lt_ck D_k), c > k, and
    // eliminates column k from the inverse, N_r. -= Lt_rk N_k. (row k of N is final by then).  No LDS access and no
    // sqrt / divide inside the chain (one v_rcp_f64 + two Newton steps per pivot); L = Lt sqrt(D), N = sqrt(D)^-1 Lt^-1
    // afterwards.  scripts/micro/diag32.hip measured the alternatives on one 32 x 32 block: this form 11.7 us, rows in
    // registers + inverse through LDS 22.8 us, 256 threads with everything in LDS and workgroup barriers 19.6 us,
    // one wavefront with everything in LDS 49 us.
    // The two halves of every step run in TWO wavefronts: this one factorises and hands each finished column of Lt to
    // wavefront 5 through LDS (column + a counter; LDS operations of one wavefront complete in order), which eliminates it
    // from the inverse one step behind (scripts/micro/diag32.hip: 12.0 -> 9.8 us).
    for (int e = lane; e < 1024; e += 64) Dm[(e >> 5) * 33 + (e & 31)] = dwork[(int64_t)kb * 1024 + e];
    wave_lds_sync();
    const int row = lane & 31;
    double a[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) a[c] = Dm[row * 33 + c];
    double dsel = 1.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const double d = readlane_f64(a[k], k);
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      dsel = (row == k) ? d : dsel;
      const double ak = a[k];
      const double l = (row > k) ? ak * rd : 0.0;   // Lt_rk (0 on and above the diagonal: those rows are finished)
      a[k] = l;
      if (lane < 32) colL[k * 32 + row] = l;
      wave_lds_sync();
      if (lane == 0) __hip_atomic_store(pflag, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int c = k + 1; c < 32; ++c) a[c] -= l * readlane_f64(ak, c);
    }
    const double sd = sqrt(dsel), isd = 1.0 / sd;
    if (lane < 32) {
      colL[32 * 32 + row] = isd;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const double sc = readlane_f64(sd, c);
        Dm[row * 33 + c] = (c < row) ? a[c] * sc : ((c == row) ? sd : 0.0);
      }
    }
    wave_lds_sync();
    if (lane == 0) __hip_atomic_store(pflag, 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else if (w == 5) {
    // ---- the inverse of the unit triangle, one step behind the factorisation: N_r. -= Lt_rk N_k. (row k of N is final
    // by then), rows in registers, v_readlane broadcasts; N = sqrt(D)^-1 Lt^-1 at the end.  The waits are bounded: a lost
    // hand-over would give wrong numbers (the tests see them), never a hang
    const int row = lane & 31;
    double n[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) n[c] = (c == row) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      int spins = 0;
      while (__hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k && ++spins < (1 << 22)) {}
      wave_lds_sync();
      const double l = colL[k * 32 + row];
#pragma unroll
      for (int c = 0; c <= k; ++c) n[c] -= l * readlane_f64(n[c], k);
    }
    int spins = 0;
    while (__hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 33 && ++spins < (1 << 22)) {}
    wave_lds_sync();
    const double isd = colL[32 * 32 + row];
    if (lane < 32) {
#pragma unroll
      for (int c = 0; c < 32; ++c) Li[row * 33 + c] = n[c] * isd;
    }
  } else if (has_work) {
    // ---- accumulation, j dealt to the four wavefronts
    double* At = tiles + (size_t)w * 2 * CHOL_T;
    double* Bt = At + CHOL_T;
    const int mr = lane & 15, mq = lane >> 4;
    v4f64 m00 = {0.0, 0.0, 0.0, 0.0}, m01 = m00, m10 = m00, m11 = m00;
    const int j_lo = type1 ? kcol : 0;
    // rows of the A tile: block row i (type 0) or kb (type 1) of L; B tile: block (kb, j) of L, transposed use (type 0),
    // or block (j, kcol) of N, plain use (type 1)
    const double* Abase = Cm + (int64_t)(32 * (type1 ? kb : i)) * ld;
    double2 ra[8], rb[8];
    auto request = [&](int j) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int e = it * 128 + lane * 2;
        const int r = e >> 5, k = e & 31;
        ra[it] = *reinterpret_cast<const double2*>(Abase + (int64_t)r * ld + 32 * j + k);
        rb[it] = type1 ? *reinterpret_cast<const double2*>(Nm + (int64_t)(32 * j + r) * ld + 32 * kcol + k)
                       : *reinterpret_cast<const double2*>(Cm + (int64_t)(32 * kb + r) * ld + 32 * j + k);
      }
    };
    int j = j_lo + w;
    if (j < kb) request(j);
    for (; j < kb; j += 4) {
      wave_lds_sync();   // the previous tiles have been read
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int e = it * 128 + lane * 2;
        const int r = e >> 5, k = e & 31;
        At[k * 33 + r] = ra[it].x;          // At[t][r] = A[r][t]
        At[(k + 1) * 33 + r] = ra[it].y;
        if (type1) {                        // Bt[t][c] = N_jk[t][c]: the tile as it is
          Bt[r * 33 + k] = rb[it].x;
          Bt[r * 33 + k + 1] = rb[it].y;
        } else {                            // Bt[t][c] = L_kj[c][t]
          Bt[k * 33 + r] = rb[it].x;
          Bt[(k + 1) * 33 + r] = rb[it].y;
        }
      }
      wave_lds_sync();
      if (j + 4 < kb) request(j + 4);
      // 32 x 32 x 32 product on the matrix cores: v_mfma_f64_16x16x4_f64, 2 x 2 tiles of 16 x 16, 8 steps of k = 4.  Operand
      // layout (MI355X_MICROARCH.md): A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], one f64 per
      // lane each; C/D: 4 f64 per lane, col = lane & 15, row = (lane >> 4) + 4 reg.  The LDS tiles are k-major, so every
      // operand is one conflict-free read; against 4 x 4 register tiles on the vector pipe (8 LDS reads per 16 FMAs, LDS
      // bound with four wavefronts per compute unit) the product needs 1/8 of the LDS reads
#pragma unroll
      for (int k0 = 0; k0 < 32; k0 += 4) {
        const double a0 = At[(k0 + mq) * 33 + mr], a1 = At[(k0 + mq) * 33 + 16 + mr];
        const double b0 = Bt[(k0 + mq) * 33 + mr], b1 = Bt[(k0 + mq) * 33 + 16 + mr];
        m00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, m00, 0, 0, 0);
        m01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, m01, 0, 0, 0);
        m10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, m10, 0, 0, 0);
        m11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, m11, 0, 0, 0);
      }
    }
    wave_lds_sync();
    // partial sums -> LDS (the wave's own A tile, now as [r][c]): tile (rt, ct), register g -> row 16 rt + mq + 4 g, col 16 ct + mr
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      At[(mq + 4 * g) * 33 + mr] = m00[g];
      At[(mq + 4 * g) * 33 + 16 + mr] = m01[g];
      At[(16 + mq + 4 * g) * 33 + mr] = m10[g];
      At[(16 + mq + 4 * g) * 33 + 16 + mr] = m11[g];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && tid < 256) {
    for (int e = tid; e < 1024; e += 256) {
      const int r = e >> 5, c = e & 31;
      Cm[(int64_t)(32 * kb + r) * ld + 32 * kb + c] = c <= r ? Dm[r * 33 + c] : 0.0;
      Nm[(int64_t)(32 * kb + r) * ld + 32 * kb + c] = Li[r * 33 + c];
    }
  }
  if (!has_work) return;
  double* Ps = tiles + CHOL_T;   // wave 0's B tile: P, later the result
  double pv[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < 256) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int o = pr * 33 + pc + v;
      const double sum = ((tiles[o] + tiles[2 * CHOL_T + o]) + tiles[4 * CHOL_T + o]) + tiles[6 * CHOL_T + o];
      pv[v] = type1 ? sum : c_own[v] - sum;
    }
  }
  __syncthreads();
  if (tid < 256) {
#pragma unroll
    for (int v = 0; v < 4; ++v) Ps[pr * 33 + pc + v] = pv[v];
  }
  __syncthreads();
  double xv[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < 256) {
    if (type1) {  // N_kb,k = -Linv P:  out[r][c] = -sum_t Li[r][t] P[t][c]
#pragma unroll 8
      for (int k = 0; k < 32; ++k) {
        const double lk = Li[pr * 33 + k];
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[v] -= lk * Ps[k * 33 + pc + v];
      }
    } else {      // L_ik = P Linv':  out[r][c] = sum_t P[r][t] Li[c][t]
#pragma unroll 8
      for (int k = 0; k < 32; ++k) {
        const double pk = Ps[pr * 33 + k];
#pragma unroll
        for (int v = 0; v < 4; ++v) xv[v] += pk * Li[(pc + v) * 33 + k];
      }
    }
  }
  __syncthreads();
  if (tid < 256) {
    double* out = type1 ? Nm + (int64_t)(32 * kb + pr) * ld + 32 * kcol + pc : Cm + (int64_t)(32 * i + pr) * ld + 32 * kb + pc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      out[v] = xv[v];
      Ps[pr * 33 + pc + v] = xv[v];
    }
  }
  if (type1) return;
  __syncthreads();
  if (tid < 256) {  // dwork_i -= L_ik L_ik'
    double dv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double xk = Ps[pr * 33 + k];
#pragma unroll
      for (int v = 0; v < 4; ++v) dv[v] += xk * Ps[(pc + v) * 33 + k];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) dwork[(int64_t)i * 1024 + pr * 32 + pc + v] = d_own[v] - dv[v];
  }
}

// y = N x (trans = 0: one workgroup per block row) or y = N' x (trans = 1: one workgroup per block column), N lower
// block-triangular with full 32 x 32 blocks (zeros above the diagonal inside the diagonal blocks)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_tri_apply(const double* __restrict__ Nm, int ld, int nb, const double* __restrict__ x,
                                                    double* __restrict__ y, int trans) {
  __shared__ double red[8][32];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  if (!trans) {
    const int r = tid >> 3, p = tid & 7;
    const double* row = Nm + (int64_t)(32 * b + r) * ld;
    double s = 0.0;
#pragma unroll 8
    for (int c = p; c < 32 * (b + 1); c += 8) s += row[c] * x[c];
    s += __shfl_xor(s, 1, 8);
    s += __shfl_xor(s, 2, 8);
    s += __shfl_xor(s, 4, 8);
    if (p == 0) y[32 * b + r] = s;
  } else {
    const int c = tid & 31, p = tid >> 5;
    double s = 0.0;
#pragma unroll 8
    for (int r = 32 * b + p; r < 32 * nb; r += 8) s += Nm[(int64_t)r * ld + 32 * b + c] * x[r];
    red[p][c] = s;
    __syncthreads();
    if (p == 0) {
      double t = red[0][c];
#pragma unroll
      for (int q = 1; q < 8; ++q) t += red[q][c];
      y[32 * b + c] = t;
    }
  }
}

// w[p] = (row p of V) . x  for a single vector x given as column `xcol` of X ([3n][xld]); rows K .. Kp-1 = 0
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_dlr_vdot(DlrArgs A, const double* __restrict__ X, int xld, int xcol, double* __restrict__ w) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= A.Kp) return;
  double val = 0.0;
  if (p < A.K) {
    const int j = p / 3, k = p - 3 * j;
    const int a = A.va[j], b = A.vb[j];
    const double* v = A.vrec + (int64_t)j * DLR_V + 3 * k;
    const double* xa = X + (3 * (int64_t)a) * xld + xcol;
    const double* xb = X + (3 * (int64_t)b) * xld + xcol;
    val = v[0] * xa[0] + v[1] * xa[xld] + v[2] * xa[2 * (int64_t)xld] + v[9] * xb[0] + v[10] * xb[xld] + v[11] * xb[2 * (int64_t)xld];
  }
  w[p] = val;
}

// y[row] (+)= t[row] - sum_q Z[row][q] w[q]; one wavefront per row, t = column tcol of Tm ([rows][tld])
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_dlr_combine(const double* __restrict__ Z, int ld, int K, const double* __restrict__ w,
                                                      const double* __restrict__ Tm, int tld, int tcol, int nrows,
                                                      double* __restrict__ y, int add) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= nrows) return;
  double s = 0.0;
  const double* zr = Z + (int64_t)row * ld;
  for (int q = lane; q < K; q += 64) s += zr[q] * w[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) {
    const double v = Tm[(int64_t)row * tld + tcol] - s;
    y[row] = add ? y[row] + v : v;
  }
}

// r = b - Ap
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_dlr_resid(int64_t n, const double* __restrict__ b, const double* __restrict__ ap, double* __restrict__ r) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) r[i] = b[i] - ap[i];
}

}  // namespace dev
}  // namespace pgo
