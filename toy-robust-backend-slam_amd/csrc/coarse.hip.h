// Second level of the PCG preconditioner: an additive coarse correction on RIGID-BODY MODES of pose aggregates.
//
//     M^-1  =  M1^-1  +  P (P' (H + D'D) P)^-1 P'
//
// M1 is the one-level block preconditioner (3x3 pose blocks, dense pose groups or chain segments, kernels.hip.h).  It
// removes the error inside a block and nothing of the smooth, long-range error that dominates these systems once the
// trust region has opened up -- which is why block-Jacobi PCG needs thousands of iterations for the exact solve the
// reference's SPARSE_NORMAL_CHOLESKY does (main.cpp:154-163).  The coarse space: aggregates of A consecutive
// poses (multiples of M1's blocks), three unknowns per aggregate = the rigid motions of the aggregate as a whole,
//     translation x, translation y, rotation about the aggregate's centre c:
//         d pose_i  =  B_i q,    B_i = [ 1 0 -(y_i - c_y) ; 0 1 (x_i - c_x) ; 0 0 1 ]
// Every relative-pose residual (src/ceres_error.cpp:42-94) is invariant under a rigid motion of both endpoints, so these
// are exactly the near-null vectors of J'J inside an aggregate; piecewise CONSTANT pose offsets (the first thing one tries)
// are not -- a constant heading offset without the matching displacement is a high-energy mode (measured on M3500,
// METHOD 1, LM iteration 3, PCG to 1e-10: one level 1557 iterations, constant basis 705, rigid-body basis 178).
// In the Jacobi-scaled variables of the linear system (step = S y) the basis is P_i = S_i^-1 B_i: five numbers per pose
// (the planes `pb`); the constant pose has S = 0 and takes no part.
//
// Per LM iteration: centres + basis planes (k_coarse_basis), the Galerkin matrix P'(H + D'D)P, dense, order 3 x aggregates,
// every 3x3 block summed by one wavefront in a fixed order (k_coarse_assemble: no atomics, bitwise reproducible), its
// Cholesky factorisation with the explicit inverse factor N = L^-1 (k_chol_panel of direct.hip.h, fp64 matrix cores).
// Per PCG iteration: r_c = P'r (k_coarse_restrict); e_c = A_c^-1 r_c -- one product with the explicit inverse N'N where the
// coarse order is small (k_coarse_matvec, order <= 1024), else the two triangular products N'(N r_c) (k_tri_apply) -- with
// the partials of r_c . e_c, the coarse level's share of r.z; and the prolongation folded into the direction update,
// p = (z + P e_c) + beta p (k_cg_update2c).  A factorisation that lost positive definiteness to rounding switches the
// level off for that LM iteration (k_coarse_check: a device flag every apply kernel reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hip.h"

namespace pgo {
namespace dev {

struct CoarseArgs {
  int32_t n_loc;       // poses (one rank: the coarse level is built for world == 1)
  int32_t agg;         // poses per aggregate
  int32_t n_agg;       // aggregates
  int32_t K, Kp;       // 3 n_agg, rounded up to a multiple of 32 (order of the padded coarse matrix)
  const double* poses; // [n x 3]
  const double* scale; // [n x 3] Jacobi column scales (0 on constant poses)
  double* pb;          // 5 planes [n_loc]: 1/s0, 1/s1, 1/s2, -(y - cy)/s0, (x - cx)/s1
  // fine matrix
  const double* hoff;  // off-diagonal blocks (AoSoA, hoff_index)
  const double* hd;    // 6 planes: diagonal blocks
  const double* d2;    // [n x 3] LM diagonal
  const int32_t* inc_col;
  // coarse blocks with at least one fine entry: block b couples aggregates (cb_i[b], cb_j[b]); its off-diagonal fine blocks are
  // the incidences cb_q[cb_ptr[b] .. cb_ptr[b+1]) (ascending), with rows cb_row[]
  const int32_t* cb_i;
  const int32_t* cb_j;
  const int32_t* cb_ptr;
  const int32_t* cb_q;
  const int32_t* cb_row;
  int32_t n_cb;
  double* cap;         // [Kp][Kp] coarse matrix (zeroed by the caller before k_coarse_assemble)
  double* dwork;       // [Kp / 32][32][32] its diagonal blocks once more (k_chol_panel's input)
};

// the five numbers of P_i = S_i^-1 B_i
struct PBasis {
  double a0, a1, a2, b0, b1;
};
__device__ __forceinline__ PBasis pb_load(const double* __restrict__ pb, int64_t n, int64_t i) {
  PBasis p;
  p.a0 = pb[i];
  p.a1 = pb[n + i];
  p.a2 = pb[2 * n + i];
  p.b0 = pb[3 * n + i];
  p.b1 = pb[4 * n + i];
  return p;
}

// sum over the wavefront in a fixed tree, result in lane 0
__device__ __forceinline__ double wave_sum_fixed(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// one wavefront per aggregate: centre of the aggregate's poses, then the basis planes of its poses
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_basis(CoarseArgs A) {
  const int w = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (w >= A.n_agg) return;
  const int i0 = w * A.agg, i1 = min(A.n_loc, i0 + A.agg);
  double sx = 0.0, sy = 0.0;
  for (int i = i0 + lane; i < i1; i += 64) {
    sx += A.poses[3 * (int64_t)i];
    sy += A.poses[3 * (int64_t)i + 1];
  }
  sx = __shfl(wave_sum_fixed(sx), 0, 64);
  sy = __shfl(wave_sum_fixed(sy), 0, 64);
  const double cx = sx / (double)(i1 - i0), cy = sy / (double)(i1 - i0);
  const int64_t n = A.n_loc;
  for (int i = i0 + lane; i < i1; i += 64) {
    const double s0 = A.scale[3 * (int64_t)i], s1 = A.scale[3 * (int64_t)i + 1], s2 = A.scale[3 * (int64_t)i + 2];
    // a pose without any edge (an all-zero row of J'J) stays out of the coarse space, like the constant pose: its row of
    // the system is D'D y = 0, and a correction through its aggregate would move it by rounding noise
    const bool in_graph = A.hd[i] != 0.0 || A.hd[3 * n + i] != 0.0 || A.hd[5 * n + i] != 0.0;
    const double a0 = (in_graph && s0 > 0.0) ? 1.0 / s0 : 0.0, a1 = (in_graph && s1 > 0.0) ? 1.0 / s1 : 0.0,
                 a2 = (in_graph && s2 > 0.0) ? 1.0 / s2 : 0.0;
    A.pb[i] = a0;
    A.pb[n + i] = a1;
    A.pb[2 * n + i] = a2;
    A.pb[3 * n + i] = -(A.poses[3 * (int64_t)i + 1] - cy) * a0;
    A.pb[4 * n + i] = (A.poses[3 * (int64_t)i] - cx) * a1;
  }
}

// C += P_i' H P_j for a general 3x3 H (row-major)
__device__ __forceinline__ void coarse_accumulate(double (&C)[9], const PBasis& pi, const double (&h)[9], const PBasis& pj) {
  double T[9];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    T[3 * k] = h[3 * k] * pj.a0;
    T[3 * k + 1] = h[3 * k + 1] * pj.a1;
    T[3 * k + 2] = h[3 * k] * pj.b0 + h[3 * k + 1] * pj.b1 + h[3 * k + 2] * pj.a2;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    C[c] += pi.a0 * T[c];
    C[3 + c] += pi.a1 * T[3 + c];
    C[6 + c] += pi.b0 * T[c] + pi.b1 * T[3 + c] + pi.a2 * T[6 + c];
  }
}

// One wavefront per coarse block (I, J): the lanes take the block's fine entries l, l + 64, ... in list order, the 64
// partial 3x3 sums are added in a fixed tree -- the same bits on every run.  Block (I, I) also takes the diagonal blocks
// H_ii + D'D of its poses.  Writes the block into the dense matrix and, for blocks on the block diagonal of the 32 x 32
// partition, into `dwork`.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_assemble(CoarseArgs A) {
  const int b = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (b >= A.n_cb) return;
  const int I = A.cb_i[b], J = A.cb_j[b];
  const int64_t n = A.n_loc;
  double C[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int k = A.cb_ptr[b] + lane; k < A.cb_ptr[b + 1]; k += 64) {
    const int q = A.cb_q[k], row = A.cb_row[k];
    const int col = A.inc_col[q];
    double h[9];
    hoff_load(A.hoff, q, h);
    coarse_accumulate(C, pb_load(A.pb, n, row), h, pb_load(A.pb, n, col));
  }
  if (I == J) {
    const int i0 = I * A.agg, i1 = min(A.n_loc, i0 + A.agg);
    for (int i = i0 + lane; i < i1; i += 64) {
      const double d00 = A.hd[i] + A.d2[3 * (int64_t)i], d01 = A.hd[n + i], d02 = A.hd[2 * n + i];
      const double d11 = A.hd[3 * n + i] + A.d2[3 * (int64_t)i + 1], d12 = A.hd[4 * n + i], d22 = A.hd[5 * n + i] + A.d2[3 * (int64_t)i + 2];
      const double h[9] = {d00, d01, d02, d01, d11, d12, d02, d12, d22};
      const PBasis p = pb_load(A.pb, n, i);
      coarse_accumulate(C, p, h, p);
    }
  }
#pragma unroll
  for (int c = 0; c < 9; ++c) C[c] = wave_sum_fixed(C[c]);
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int p = 3 * I + m, q = 3 * J + c;
        A.cap[(int64_t)p * A.Kp + q] = C[3 * m + c];
        if ((p >> 5) == (q >> 5)) A.dwork[(int64_t)(p >> 5) * 1024 + (p & 31) * 32 + (q & 31)] = C[3 * m + c];
      }
  }
}

// identity on the padding rows K .. Kp-1 of the (zeroed) coarse matrix and of dwork
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_coarse_pad(double* __restrict__ cap, double* __restrict__ dwork, int K, int Kp) {
  const int p = K + blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Kp) return;
  cap[(int64_t)p * Kp + p] = 1.0;
  dwork[(int64_t)(p >> 5) * 1024 + (p & 31) * 32 + (p & 31)] = 1.0;
}

// r_c = P' r: one wavefront per aggregate, fixed order
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_restrict(int n_loc, int agg, int n_agg, const double* __restrict__ pb,
                                                         const double* __restrict__ r, double* __restrict__ rc,
                                                         const int32_t* __restrict__ done) {
  if (done && *done) return;
  const int w = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (w >= n_agg) return;
  const int i0 = w * agg, i1 = min(n_loc, i0 + agg);
  const int64_t n = n_loc;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0;
  for (int i = i0 + lane; i < i1; i += 64) {
    const PBasis p = pb_load(pb, n, i);
    const double r0 = r[3 * (int64_t)i], r1 = r[3 * (int64_t)i + 1], r2 = r[3 * (int64_t)i + 2];
    c0 += p.a0 * r0;
    c1 += p.a1 * r1;
    c2 += p.b0 * r0 + p.b1 * r1 + p.a2 * r2;
  }
  c0 = wave_sum_fixed(c0);
  c1 = wave_sum_fixed(c1);
  c2 = wave_sum_fixed(c2);
  if (lane == 0) {
    rc[3 * w] = c0;
    rc[3 * w + 1] = c1;
    rc[3 * w + 2] = c2;
  }
}

// z += P e_c and the same into the gather vector p (the PCG start-up, where p = z); skipped as a whole when the level is off
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_prolong(int n_loc, int agg, const double* __restrict__ pb, const double* __restrict__ ec,
                                                        double* __restrict__ z, double* __restrict__ p, const int32_t* __restrict__ ok) {
  if (!*ok) return;
  const int64_t n = n_loc;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_loc; i += gridDim.x * 256) {
    const PBasis b = pb_load(pb, n, i);
    const int I = i / agg;
    const double e0 = ec[3 * I], e1 = ec[3 * I + 1], e2 = ec[3 * I + 2];
    const double z0 = b.a0 * e0 + b.b0 * e2, z1 = b.a1 * e1 + b.b1 * e2, z2 = b.a2 * e2;
    double* zz = z + 3 * (int64_t)i;
    zz[0] += z0;
    zz[1] += z1;
    zz[2] += z2;
    if (p) {
      double* pp = p + PS * (int64_t)i;
      pp[0] += z0;
      pp[1] += z1;
      pp[2] += z2;
    }
  }
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_fill(double* __restrict__ x, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// explicit inverse of the coarse matrix, Ainv = N'N (N = L^-1, lower triangular), for small orders: one thread per entry
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_ainv(const double* __restrict__ Nm, int Kp, double* __restrict__ Ainv) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= Kp) return;
  double s = 0.0;
  for (int k = max(i, j); k < Kp; ++k) s += Nm[(int64_t)k * Kp + i] * Nm[(int64_t)k * Kp + j];
  Ainv[(int64_t)i * Kp + j] = s;
}

// the level is usable iff a probe of the factor is finite: x = Ainv 1 (explicit inverse) or x = N'(N 1) computed by the
// caller with k_tri_apply; one workgroup
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(256) void k_coarse_check(const double* __restrict__ x, int K, int32_t* __restrict__ ok) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int nf = 0;
  for (int k = threadIdx.x; k < K; k += 256) nf |= !isfinite(x[k]);
  if (nf) bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) *ok = bad ? 0 : 1;
}

// e_c = Ainv r_c (dense, symmetric, order Kp <= 1024): one wavefront per row, four rows per workgroup; the workgroup's
// share of r_c . e_c goes to dot_part[blockIdx.x] (fixed order).  Level off: e_c = 0, partials 0.
// (Folding the restriction into this kernel's prologue -- every workgroup forming r_c in LDS, one thread per aggregate --
// saves a launch and was measured slower: M3500 METHOD 1 113 against 156 GN it/s.)
template <int PGO_UNIT_ = 0>
__global__ __launch_bounds__(256) void k_coarse_matvec(const double* __restrict__ Ainv, int Kp, const double* __restrict__ rc,
                                                       double* __restrict__ ec, double* __restrict__ dot_part,
                                                       const int32_t* __restrict__ ok, const int32_t* __restrict__ done) {
  __shared__ double sh[4];
  if (done && *done) return;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  double s = 0.0;
  if (*ok && row < Kp) {
    const double* a = Ainv + (int64_t)row * Kp;
    for (int c = lane; c < Kp; c += 64) s += a[c] * rc[c];
  }
  s = wave_sum_fixed(s);
  if (lane == 0) {
    if (row < Kp) ec[row] = s;
    sh[w] = (row < Kp) ? s * rc[row] : 0.0;
  }
  __syncthreads();
  if (threadIdx.x == 0) dot_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// larger orders, after the two triangular products N'(N r_c) (k_tri_apply): partials of r_c . e_c, 256 entries per
// workgroup; level off: e_c = 0.  (Cutting the two products into 32 x 512 tiles over a few hundred workgroups was measured
// equal -- 100k poses, order 4689: 12.2 against 12.3 GN it/s -- the products are bound by reading the 88 MB factor twice.)
template <int PGO_UNIT_ = 0>
__global__ __launch_bounds__(256) void k_coarse_dot(int Kp, const double* __restrict__ rc, double* __restrict__ ec,
                                                    double* __restrict__ dot_part, const int32_t* __restrict__ ok,
                                                    const int32_t* __restrict__ done) {
  __shared__ double red[8];
  if (done && *done) return;
  const int k = blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (k < Kp) {
    if (!*ok) ec[k] = 0.0;
    else v = rc[k] * ec[k];
  }
  v = block_sum_bcast(v, red);
  if (threadIdx.x == 0) dot_part[blockIdx.x] = v;
}

// k_cg_update2 with the prolongation folded in: beta = rz_new / rz ; p = (z + P e_c) + beta p, where rz_new sums the
// one-level partials AND the coarse level's partials of r_c . e_c (n_rz covers both).  Workgroup 0 publishes the scalars.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_update2c(CgVec V, int parity, const double* __restrict__ part_rz, int n_rz,
                                                    const double* __restrict__ part_rr, int n_rr, int agg,
                                                    const double* __restrict__ pb, const double* __restrict__ ec) {
  __shared__ double red[8];
  if (V.st->done) return;
  const double rz_new = sum_partials_bcast(part_rz, n_rz, red);
  const double rr = sum_partials_bcast(part_rr, n_rr, red);
  const double rz_old = V.st->rz[parity];
  const double tol2 = V.st->tol2;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    V.st->rz[parity ^ 1] = rz_new;
    V.st->rr = rr;
    V.st->iters += 1;
  }
  if (rr <= tol2) {  // converged: leave p alone, freeze the solve (same decision in every workgroup)
    if (blockIdx.x == 0 && threadIdx.x == 0) V.st->done = 1;
    return;
  }
  const double beta = rz_new / rz_old;
  const int64_t n = V.n_loc;
  double* p = V.p + PS * (int64_t)V.lo;
  for (int i = blockIdx.x * WG + threadIdx.x; i < V.n_loc; i += gridDim.x * WG) {
    const PBasis b = pb_load(pb, n, i);
    const int I = i / agg;
    const double e0 = ec[3 * I], e1 = ec[3 * I + 1], e2 = ec[3 * I + 2];
    const double* zz = V.z + 3 * (int64_t)i;
    double* pp = p + PS * (int64_t)i;
    pp[0] = (zz[0] + (b.a0 * e0 + b.b0 * e2)) + beta * pp[0];
    pp[1] = (zz[1] + (b.a1 * e1 + b.b1 * e2)) + beta * pp[1];
    pp[2] = (zz[2] + b.a2 * e2) + beta * pp[2];
  }
}

}  // namespace dev
}  // namespace pgo
