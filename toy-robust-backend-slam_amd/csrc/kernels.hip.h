// HIP kernels of the pose-graph backend, written for gfx950 (CDNA4, wave64).
// Everything here is HBM-bandwidth bound fp64 work on 3x3 blocks: no MFMA.
//
//   K1  k_edge_eval      fused per-edge SE(2) residual + 3x6 Jacobian + DCS weight + Huber
//                        corrector (reference: src/ceres_error.cpp:42-94, 135-196 evaluated through
//                        AutoDiffCostFunction, main.cpp:66-68 loss); writes a 112-byte record per edge
//   K2  k_assemble       row-tiled segmented reduction of (JS)'(JS) and S J'r (S = Jacobi column
//                        scaling) into diagonal planes + one off-diagonal 3x3 block per incidence
//                        (block CSR)
//   K3  k_spmv           y = ((JS)'(JS) + D'D) p on that block CSR, fused p.y partial
//   K4  k_prepare        LM diagonal + block-Jacobi (3x3 inverse) preconditioner
//   K5  k_cg_*           fused PCG vector updates with in-kernel dot partials
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgo {
namespace dev {

constexpr int WG = 256;          // workgroup size of every kernel here (4 waves)
// Edge record, 14 doubles = 112 B (16-byte aligned):  A = d e/d P1 (9, row-major) | g2 | r (3) | cost.
// The second Jacobian block is implied by the structure of the SE(2) error (also under DCS and the
// Huber corrector, which scale whole rows / add e (x) grad psi with grad psi antisymmetric in (x,y)):
//     d e/d P2 = [ -A[:,0] | -A[:,1] | (0, 0, g2)' ]
// so storing it would only repeat 8 of its 9 entries.
constexpr int REC = 14;
constexpr int REC_LDS = 15;      // odd stride => conflict-free ds_write_b64 when staging records
// Information-weighted mode (pgo_options.info_weighting): whitening by L' (Omega = L L') mixes the rows and the chi2 form
// of DCS makes grad psi depend on theta2, so the last column of d e/d P2 is a general 3-vector b3:
//     record = A (9) | b3 (3) | r (3) | cost = 16 doubles = 128 B,   d e/d P2 = [ -A[:,0] | -A[:,1] | b3 ]
// (the error still depends on the positions only through P2 - P1, so the first two columns stay implied).
constexpr int REC_INFO = 16;
template <bool INFO> struct RecLayout {
  static constexpr int N = INFO ? REC_INFO : REC;   // doubles per record
  static constexpr int LDS = N + 1;                 // odd staging stride
  static constexpr int R0 = INFO ? 12 : 10;         // first residual entry
};
constexpr int PS = 3;            // doubles per pose in the GATHERED vector p.  Padding to 4 (32 B, never straddling a
                                 // 64-byte sector) was measured: no fewer fetched bytes (FETCH_SIZE 566 vs 557 MiB), so 3.

// ------------------------------------------------- streaming accesses
// Data that a kernel touches exactly once (matrix / factor streams, CG vectors) is moved with the non-temporal hint so
// that it does not displace what IS re-used (the gathered search direction) from the XCD's L2.
#ifndef PGO_NT_STREAMS
#define PGO_NT_STREAMS 1
#endif
template <class T>
__device__ __forceinline__ T ld_stream(const T* p) {
#if PGO_NT_STREAMS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <class T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#if PGO_NT_STREAMS
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// the same under a template switch: kernels whose phases hand data to each other through global memory INSIDE one
// workgroup (solo.hip.h) must use plain accesses (an nt load bypasses the CU's L1, a plain one may not)
template <bool NT, class T>
__device__ __forceinline__ T ld_sel(const T* p) {
  if constexpr (NT) return ld_stream(p);
  else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void st_sel(T* p, T v) {
  if constexpr (NT) st_stream(p, v);
  else *p = v;
}

// ------------------------------------------------- layout of the off-diagonal blocks
// AoSoA: incidences in groups of 64 (one wave), 9 values x 64 lanes contiguous (4608 B per group), so
// that a wave's 9 coalesced 512-byte accesses fall into ONE contiguous 4.5 KiB region instead of nine
// regions tens of MB apart (DRAM page locality; measured against plain planes).
//
// Inside a group the first eight values are stored as four (2 x 64) double2 planes and the ninth as one 64-double plane:
// a lane moves its block with four 16-byte accesses + one 8-byte access (8-byte-per-lane streams run at 0.54-0.70x the
// rate of 16-byte ones on gfx950, MI355X_MICROARCH.md "cache-policy bits"); value c of incidence q sits at hoff_index(c, q).
__device__ __forceinline__ int64_t hoff_index(int c, int64_t q) {
  const int64_t g = (q >> 6) * 576, l = q & 63;
  return c < 8 ? g + (int64_t)(c >> 1) * 128 + 2 * l + (c & 1) : g + 512 + l;
}
// the whole block of incidence q (row-major 3x3) in / out of registers
__device__ __forceinline__ void hoff_load(const double* __restrict__ hoff, int64_t q, double (&v)[9]) {
  const double* g = hoff + (q >> 6) * 576;
  const int l = (int)(q & 63);
  const double2* g2 = reinterpret_cast<const double2*>(g) + l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double2 t = g2[64 * k];
    v[2 * k] = t.x;
    v[2 * k + 1] = t.y;
  }
  v[8] = g[512 + l];
}
// the same with the non-temporal hint: the H stream is read once per product and should not displace the gathered
// search direction from L2
typedef double double2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void hoff_load_nt(const double* __restrict__ hoff, int64_t q, double (&v)[9]) {
  const double* g = hoff + (q >> 6) * 576;
  const int l = (int)(q & 63);
  const double2_v* g2 = reinterpret_cast<const double2_v*>(g) + l;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double2_v t = __builtin_nontemporal_load(g2 + 64 * k);
    v[2 * k] = t.x;
    v[2 * k + 1] = t.y;
  }
  v[8] = __builtin_nontemporal_load(g + 512 + l);
}
__device__ __forceinline__ void hoff_store(double* __restrict__ hoff, int64_t q, const double (&v)[9]) {
  double* g = hoff + (q >> 6) * 576;
  const int l = (int)(q & 63);
  double2* g2 = reinterpret_cast<double2*>(g) + l;
#pragma unroll
  for (int k = 0; k < 4; ++k) g2[64 * k] = make_double2(v[2 * k], v[2 * k + 1]);
  g[512 + l] = v[8];
}
// a pose's three doubles of the gathered vector: one 16-byte + one 8-byte load (the vector is 8-byte aligned only)
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));
template <int STRIDE = PS>
__device__ __forceinline__ void gather3(const double* __restrict__ p, int64_t col, double& p0, double& p1, double& p2) {
  const double* q = p + STRIDE * col;
  const double2_a8 t = *reinterpret_cast<const double2_a8*>(q);
  p0 = t.x;
  p1 = t.y;
  p2 = q[2];
}

// ------------------------------------------------- XCD-aware work mapping
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD; speed only, never
// correctness).  Each XCD has its own 4 MiB L2, so consecutive tiles -- which gather the same pose /
// search-direction sectors and re-read each other's edge records -- should run on ONE XCD: the item
// range is cut into 8 contiguous parts and workgroup b walks part (b % 8) with stride gridDim/8.
// Requires gridDim.x % 8 == 0.
struct XcdRange {
  int begin, end, step;
};
__device__ __forceinline__ XcdRange xcd_range(int n_items) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int chunk = (n_items + 7) >> 3;
  XcdRange r;
  r.begin = min(n_items, xcd * chunk) + slot;
  r.end = min(n_items, (xcd + 1) * chunk);
  r.step = per_xcd;
  return r;
}

// ------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
// sum over the workgroup, result broadcast to every thread.  sh: >= 5 doubles.
__device__ __forceinline__ double block_sum_bcast(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // sh may still be read from a previous call
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return sh[4];
}
__device__ __forceinline__ double block_max_bcast(double v, double* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return sh[4];
}
// every workgroup sums the same `n` partials in the same order => identical value
// everywhere, no atomics, bitwise reproducible.
__device__ __forceinline__ double sum_partials_bcast(const double* __restrict__ part, int n, double* sh) {
  double v = 0.0;
  int i = threadIdx.x;
  // (eight loads in flight per thread; the additions in the order of the plain loop)
  for (; i + 7 * WG < n; i += 8 * WG) {
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = part[i + k * WG];
#pragma unroll
    for (int k = 0; k < 8; ++k) v += a[k];
  }
  for (; i < n; i += WG) v += part[i];
  return block_sum_bcast(v, sh);
}

// ------------------------------------------------------------------- K1
struct EdgeArgs {
  const double* poses;    // [.. x 3] global pose positions
  const int32_t* ia;
  const int32_t* ib;
  const double* mx;
  const double* my;
  const double* mt;
  const uint8_t* flags;   // bit0 robust edge (DCS for METHOD 1, switchable for METHOD 2), bit1 cost counted on this rank
  int32_t n_edges;
  int32_t apply_loss;
  double phi;
  double huber_delta;
  // METHOD 2 (switchable constraints, src/ceres_error.cpp:237-317, main.cpp:115-125): e = s e_plain per robust
  // edge plus the prior sqrt(lambda) (1 - s); nullptr for METHOD 0/1
  const double* sw;       // [n_edges] switch per local edge
  double* sw_js;          // [n_edges x 3] out (with the Jacobian): d e / d s after the Huber corrector
  double sc_lambda;
  // information matrices, 6 planes [6][n_edges]: I11 I12 I13 I22 I23 I33 (include/graph.h:41-47); nullptr when the handle
  // was created without them.  Read by k_edge_eval<*, true> and k_edge_chi2 only.
  const double* info;
  // batched handles: cost of every edge (NaN where the residual or Jacobian is not finite), summed per problem by
  // k_prob_reduce; nullptr otherwise
  double* cost_out;
};

// Cholesky factor of a 3x3 information matrix, Omega = L L' (positive definiteness is checked on the host at create)
struct Chol3 {
  double l00, l10, l11, l20, l21, l22;
};
__device__ __forceinline__ Chol3 chol3(double w00, double w01, double w02, double w11, double w12, double w22) {
  Chol3 c;
  c.l00 = sqrt(w00);
  const double i0 = 1.0 / c.l00;
  c.l10 = w01 * i0;
  c.l20 = w02 * i0;
  c.l11 = sqrt(w11 - c.l10 * c.l10);
  c.l21 = (w12 - c.l20 * c.l10) / c.l11;
  c.l22 = sqrt(w22 - c.l20 * c.l20 - c.l21 * c.l21);
  return c;
}

// One lane per edge.  Algorithmic bytes per edge: 8 (ia,ib) + 24 (meas) + 1 (flags) +
// 48 (two poses) read, 112 written with the Jacobian, 0 without (INFO: + 48 read, 128 written).
template <bool WITH_JAC, bool INFO>
__global__ __launch_bounds__(WG) void k_edge_eval(EdgeArgs A, double* __restrict__ jr,
                                                  double* __restrict__ cost_part, int* __restrict__ bad) {
  constexpr int RN = RecLayout<INFO>::N, RL = RecLayout<INFO>::LDS;
  __shared__ double stage[WITH_JAC ? WG * RL : 1];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  // XCD-aware: consecutive 256-edge blocks (sorted by min endpoint => shared pose sectors) share an XCD
  const int n_blocks = (A.n_edges + WG - 1) / WG;
  const int per_xcd = (n_blocks + 7) >> 3;
  const int blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const bool live = (blockIdx.x >> 3) < per_xcd && blk < n_blocks;
  const int64_t e0 = (int64_t)blk * WG;
  const int64_t e = live ? e0 + tid : (int64_t)A.n_edges;
  double cost = 0.0;
  if (e < A.n_edges) {
    const int a = A.ia[e], b = A.ib[e];
    const double dx = A.mx[e], dy = A.my[e], dth = A.mt[e];
    const unsigned fl = A.flags[e];
    const double x1 = A.poses[3 * (int64_t)a], y1 = A.poses[3 * (int64_t)a + 1], t1 = A.poses[3 * (int64_t)a + 2];
    const double x2 = A.poses[3 * (int64_t)b], y2 = A.poses[3 * (int64_t)b + 1], t2 = A.poses[3 * (int64_t)b + 2];
    double s1, c1, s2, c2, sd, cd;
    sincos(t1, &s1, &c1);
    sincos(t2, &s2, &c2);
    sincos(dth, &sd, &cd);
    // diff = T^-1 (Ta^-1 Tb)  in closed form
    const double Dx = x2 - x1, Dy = y2 - y1;
    const double pa = c1 * Dx + s1 * Dy, pb = -s1 * Dx + c1 * Dy;  // R(t1)' D
    const double ux = pa - dx, uy = pb - dy;
    double ex = cd * ux + sd * uy, ey = -sd * ux + cd * uy;         // R(dth)' u
    const double c21 = c1 * c2 + s1 * s2, s21 = c1 * s2 - s1 * c2;  // R(t2 - t1)
    const double sind = cd * s21 - sd * c21, cosd = cd * c21 + sd * s21;
    double et = asin(sind);
    double J[18];
    if (WITH_JAC) {
      const double cm = c1 * cd - s1 * sd, sm = s1 * cd + c1 * sd;  // R(t1 + dth)
      const double g = cosd / sqrt(1.0 - sind * sind);              // d asin(u) = du / sqrt(1-u^2)
      J[0] = -cm;  J[1] = -sm;  J[2] = cd * pb - sd * pa;   J[3] = cm;   J[4] = sm;   J[5] = 0.0;
      J[6] = sm;   J[7] = -cm;  J[8] = -sd * pb - cd * pa;  J[9] = -sm;  J[10] = cm;  J[11] = 0.0;
      J[12] = 0.0; J[13] = 0.0; J[14] = -g;                 J[15] = 0.0; J[16] = 0.0; J[17] = g;
    }
    if (INFO) {  // whiten: e <- L' e, J <- L' J  (|e|^2 becomes e' Omega e)
      const int64_t ne = A.n_edges;
      const Chol3 c = chol3(A.info[e], A.info[ne + e], A.info[2 * ne + e], A.info[3 * ne + e], A.info[4 * ne + e],
                            A.info[5 * ne + e]);
      const double w0 = c.l00 * ex + c.l10 * ey + c.l20 * et, w1 = c.l11 * ey + c.l21 * et, w2 = c.l22 * et;
      ex = w0; ey = w1; et = w2;
      if (WITH_JAC) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double j0 = J[k], j1 = J[6 + k], j2 = J[12 + k];
          J[k] = c.l00 * j0 + c.l10 * j1 + c.l20 * j2;
          J[6 + k] = c.l11 * j1 + c.l21 * j2;
          J[12 + k] = c.l22 * j2;
        }
      }
    }
    const bool switchable = (fl & 1u) && A.sw != nullptr;
    double sval = 1.0, epx = 0.0, epy = 0.0, ept = 0.0;
    if (INFO && (fl & 1u)) {  // chi2 form of DCS: s = min(1, 2 phi / (phi + chi2)), chi2 = |e_w|^2, e = s e_w
      const double chi2 = ex * ex + ey * ey + et * et;
      const double sdc = 2.0 * A.phi / (A.phi + chi2);
      if (sdc < 1.0) {
        if (WITH_JAC) {
          const double k = -2.0 * sdc / (A.phi + chi2);
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const double ds = k * (ex * J[c] + ey * J[6 + c] + et * J[12 + c]);
            J[c] = sdc * J[c] + ex * ds;
            J[6 + c] = sdc * J[6 + c] + ey * ds;
            J[12 + c] = sdc * J[12 + c] + et * ds;
          }
        }
        ex *= sdc;
        ey *= sdc;
        et *= sdc;
      }
    } else if (switchable) {  // e = s e_plain ; d e / d P = s d e_plain / d P ; d e / d s = e_plain
      sval = A.sw[e];
      epx = ex; epy = ey; ept = et;
      ex *= sval; ey *= sval; et *= sval;
      if (WITH_JAC) {
#pragma unroll
        for (int c = 0; c < 18; ++c) J[c] *= sval;
      }
    } else if (fl & 1u) {  // DCS (src/ceres_error.cpp:185-193): psi = min(1, sqrt(2 phi / (phi + ex^2 + ey^2)))
      const double res = ex * ex + ey * ey;
      const double psi_org = sqrt(2.0 * A.phi / (A.phi + res));
      if (psi_org < 1.0) {
        if (WITH_JAC) {
          const double k = -psi_org / (A.phi + res);
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            const double dpsi = k * (ex * J[c] + ey * J[6 + c]);
            J[c] = psi_org * J[c] + ex * dpsi;
            J[6 + c] = psi_org * J[6 + c] + ey * dpsi;
            J[12 + c] = psi_org * J[12 + c] + et * dpsi;
          }
        }
        ex *= psi_org;
        ey *= psi_org;
        et *= psi_org;
      }
    }
    const double s = ex * ex + ey * ey + et * et;
    double rho0 = s, sc = 1.0;
    if (A.huber_delta > 0.0) {  // ceres::HuberLoss(a): b = a^2
      const double bq = A.huber_delta * A.huber_delta;
      if (s > bq) {
        const double rs = sqrt(s);
        rho0 = 2.0 * A.huber_delta * rs - bq;
        double rho1 = A.huber_delta / rs;
        rho1 = fmax(rho1, 2.2250738585072014e-308);
        if (A.apply_loss) sc = sqrt(rho1);
      }
    }
    double ecost = 0.5 * rho0;
    if (switchable) {  // SwitchPriorResidue: sqrt(lambda) (1 - s), no loss
      const double q = 1.0 - sval;
      ecost += 0.5 * A.sc_lambda * q * q;
    }
    if (fl & 2u) cost = ecost;
    bool finite = isfinite(s);
    if (WITH_JAC) {
      double* st = stage + tid * RL;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double v0 = sc * J[6 * i + 0], v1 = sc * J[6 * i + 1], v2 = sc * J[6 * i + 2];
        finite = finite && isfinite(v0) && isfinite(v1) && isfinite(v2);
        st[3 * i + 0] = v0;
        st[3 * i + 1] = v1;
        st[3 * i + 2] = v2;
      }
      if (INFO) {
        const double b0 = sc * J[5], b1 = sc * J[11], b2 = sc * J[17];
        finite = finite && isfinite(b0) && isfinite(b1) && isfinite(b2);
        st[9] = b0;
        st[10] = b1;
        st[11] = b2;
      } else {
        const double g2 = sc * J[17];
        finite = finite && isfinite(g2);
        st[9] = g2;
      }
      constexpr int R0 = RecLayout<INFO>::R0;
      st[R0] = sc * ex;
      st[R0 + 1] = sc * ey;
      st[R0 + 2] = sc * et;
      st[R0 + 3] = ecost;
      if (switchable) {
        double* js = A.sw_js + 3 * e;
        js[0] = sc * epx;
        js[1] = sc * epy;
        js[2] = sc * ept;
      }
    }
    if (!finite) atomicOr(bad, 1);
    if (A.cost_out) A.cost_out[e] = finite ? ((fl & 2u) ? ecost : 0.0) : __builtin_nan("");
  }
  if (WITH_JAC) {
    // transpose through LDS so that the 112-byte records leave as 16-byte-per-lane
    // coalesced stores (a lane-per-record store would touch ~60 lines per instruction)
    __syncthreads();
    int64_t nvalid = live ? A.n_edges - e0 : 0;
    if (nvalid > WG) nvalid = WG;
    const int ndbl = (int)nvalid * RN;
    double* out = jr + e0 * RN;  // 16-byte aligned: e0 * 112 (128)
    for (int j = tid * 2; j < ndbl; j += 2 * WG) {
      const int le = j / RN, c = j - le * RN;  // RN is even => (j, j+1) stay in one record
      double2 v;
      v.x = stage[le * RL + c];
      v.y = stage[le * RL + c + 1];
      *reinterpret_cast<double2*>(out + j) = v;
    }
  }
  const double tot = block_sum_bcast(cost, red);
  if (tid == 0) cost_part[blockIdx.x] = tot;
}

// compute_edge_mahalanobis (src/layer_manager.cpp:230-282) for every local edge that this rank counts (flags bit1):
// m = r' Omega r of the PLAIN residual r = (ex, ey, asin(clamp(sin delta))), clamped at 0, written to the caller's edge
// index.  Any symmetric Omega (no factorisation).  57 B read + 48 B information + 8 B written per edge.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_edge_chi2(EdgeArgs A, const int32_t* __restrict__ orig_edge, double* __restrict__ out) {
  const int64_t ne = A.n_edges;
  for (int64_t e = (int64_t)blockIdx.x * WG + threadIdx.x; e < ne; e += (int64_t)gridDim.x * WG) {
    if (!(A.flags[e] & 2u)) continue;
    const int a = A.ia[e], b = A.ib[e];
    const double dx = A.mx[e], dy = A.my[e], dth = A.mt[e];
    const double x1 = A.poses[3 * (int64_t)a], y1 = A.poses[3 * (int64_t)a + 1], t1 = A.poses[3 * (int64_t)a + 2];
    const double x2 = A.poses[3 * (int64_t)b], y2 = A.poses[3 * (int64_t)b + 1], t2 = A.poses[3 * (int64_t)b + 2];
    double s1, c1, s2, c2, sd, cd;
    sincos(t1, &s1, &c1);
    sincos(t2, &s2, &c2);
    sincos(dth, &sd, &cd);
    const double Dx = x2 - x1, Dy = y2 - y1;
    const double pa = c1 * Dx + s1 * Dy, pb = -s1 * Dx + c1 * Dy;
    const double ux = pa - dx, uy = pb - dy;
    const double ex = cd * ux + sd * uy, ey = -sd * ux + cd * uy;
    const double c21 = c1 * c2 + s1 * s2, s21 = c1 * s2 - s1 * c2;
    const double sind = cd * s21 - sd * c21;
    const double et = asin(fmin(1.0, fmax(-1.0, sind)));
    const double w00 = A.info[e], w01 = A.info[ne + e], w02 = A.info[2 * ne + e], w11 = A.info[3 * ne + e],
                 w12 = A.info[4 * ne + e], w22 = A.info[5 * ne + e];
    const double m = ex * (w00 * ex + w01 * ey + w02 * et) + ey * (w01 * ex + w11 * ey + w12 * et) +
                     et * (w02 * ex + w12 * ey + w22 * et);
    out[orig_edge[e]] = m < 0.0 ? 0.0 : m;
  }
}

// input record of the chain factorisation, one 128-byte line per row: M_ii = H_ii + D'D (6: 00 01 02 11 12 22) | C_i = the
// sum of the blocks (i, i-1) of row i, row-major (9; 0 at a segment start) | pad
constexpr int CHAIN_REC = 16;

// ------------------------------------------------------------------- K2
// local row (in [r0, r1)) whose incidence range contains q: binary search in inc_ptr (<= 8 steps, L1/L2 hits)
__device__ __forceinline__ int upper_row(const int32_t* __restrict__ inc_ptr, int r0, int r1, int q) {
  int lo = r0, hi = r1 - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (inc_ptr[mid] <= q) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

struct AsmArgs {
  const double* jr;          // edge records
  const int32_t* inc_ptr;    // n_loc + 1
  const int32_t* inc_edge;   // (local edge << 1) | side
  const int32_t* inc_col;    // global pose position of the other endpoint
  const int32_t* tile_row;   // n_tiles + 1 (local rows)
  const uint8_t* inc_rowoff; // row of the incidence - first row of its tile
  const int4* tile_desc;     // per tile {first local row, rows, first incidence, incidences} (shared with K3)
  const double* scale;       // [.. x 3] Jacobi column scales, global pose positions (0 on the constant pose)
  int32_t n_tiles;
  int32_t n_loc;
  int32_t lo;                // first owned global row
  int64_t inc_stride;        // plane stride of hoff (>= n_inc)
  double* hoff;              // [(n_inc+63)/64][9][64]: (J_self)'(J_other), row-major 3x3, see hoff_index
  double* hd;                // 6 planes [n_loc]: d00 d01 d02 d11 d12 d22 of (J_self)'(J_self) summed
  double* gs;                // [n_loc x 3]: sum (J_self)' r
  // METHOD 2 only (k_assemble<true>): per-edge elimination coefficients of the switches (see k_switch_prepare) and the
  // UNREDUCED diagonal / gradient, which Ceres' LM diagonal and gradient tolerance are defined on
  const double* sw_js;       // [edges x 3]
  const double* sw_c;
  const double* sw_gamma;
  double* diag_full;         // [n_loc x 3]
  double* gs_full;           // [n_loc x 3]
  // chain preconditioner: input records of the factorisation (CHAIN_REC doubles per row, C part written here); nullptr = off
  double* chain_rec;
  int32_t chain_seg;
};

// One workgroup per tile of rows.  Phase A: one lane per incidence reads the
// 112-byte edge record (7 x 16-byte loads), expands the implied second block, applies the Jacobi column
// scales of both endpoints, forms the off-diagonal block (stored
// straight to its plane slot, coalesced) and its diagonal/gradient contribution
// (9 doubles, staged in LDS).  Phase B: one thread per (row, component) sums its
// row's staged contributions in incidence order -- a fixed order, so the result is
// bitwise reproducible and independent of the sharding.
// Everything one incidence contributes: the off-diagonal block (stored), the chain record's C part, and the nine (SC:
// fifteen) staged values of its row's diagonal block / gradient.  ed = (local edge << 1) | side.
template <bool SC, bool INFO>
__device__ __forceinline__ void asm_incidence(const AsmArgs& A, int q, int ed, int64_t col, int row, bool first_of_pair, int tid,
                                              double (*scr)[WG]) {
  constexpr int RN = RecLayout<INFO>::N, R0 = RecLayout<INFO>::R0;
  const double2* rp = reinterpret_cast<const double2*>(A.jr + (int64_t)(ed >> 1) * RN);
  double R[RN];
#pragma unroll
  for (int k = 0; k < RN / 2; ++k) {
    const double2 v = rp[k];
    R[2 * k] = v.x;
    R[2 * k + 1] = v.y;
  }
  // Jacobi column scales of this row's pose and of the other endpoint
  double ss[3], so[3];
  {
    const double* s_self = A.scale + 3 * (int64_t)(A.lo + row);
    const double* s_oth = A.scale + 3 * col;
    ss[0] = s_self[0]; ss[1] = s_self[1]; ss[2] = s_self[2];
    so[0] = s_oth[0]; so[1] = s_oth[1]; so[2] = s_oth[2];
  }
  // S[k*3+a] = d e_k / d (self pose)_a * scale, O likewise for the other endpoint.
  // side 0: self = P1 (block A), other = P2 (implied block); side 1: the reverse.
  double S[9], O[9];
  const bool self_is_a = (ed & 1) == 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a0 = R[3 * k], a1 = R[3 * k + 1], a2 = R[3 * k + 2];
    const double b2 = INFO ? R[9 + k] : ((k == 2) ? R[9] : 0.0);
    const double x0 = self_is_a ? a0 : -a0, x1 = self_is_a ? a1 : -a1, x2 = self_is_a ? a2 : b2;
    const double y0 = self_is_a ? -a0 : a0, y1 = self_is_a ? -a1 : a1, y2 = self_is_a ? b2 : a2;
    S[3 * k] = x0 * ss[0]; S[3 * k + 1] = x1 * ss[1]; S[3 * k + 2] = x2 * ss[2];
    O[3 * k] = y0 * so[0]; O[3 * k + 1] = y1 * so[1]; O[3 * k + 2] = y2 * so[2];
  }
  // METHOD 2: elimination coefficients of this edge's switch (c == 0 for ordinary edges)
  double cc = 0.0, gam = 0.0, vs[3] = {0.0, 0.0, 0.0}, vo[3] = {0.0, 0.0, 0.0};
  if constexpr (SC) {
    const int64_t le = ed >> 1;
    cc = A.sw_c[le];
    if (cc != 0.0) {
      const double* j = A.sw_js + 3 * le;
      gam = A.sw_gamma[le];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        vs[a] = S[a] * j[0] + S[3 + a] * j[1] + S[6 + a] * j[2];
        vo[a] = O[a] * j[0] + O[3 + a] * j[1] + O[6 + a] * j[2];
      }
    }
  }
  double HB[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double v = S[a] * O[b] + S[3 + a] * O[3 + b] + S[6 + a] * O[6 + b];
      if constexpr (SC) v -= cc * vs[a] * vo[b];  // J'(I - c j j')J
      HB[3 * a + b] = v;
    }
  hoff_store(A.hoff, q, HB);
  // chain preconditioner: the block (i, i-1) goes straight into the factorisation's input record (C part), so that
  // no kernel has to dig it out of the AoSoA block stream again (k_chain_extract read 975 MB to find 72 MB).  With
  // several edges between i-1 and i the first incidence writes and k_chain_dupfix replaces it by the ordered sum.
  if (A.chain_rec != nullptr && col == (int64_t)A.lo + row - 1 && (row % A.chain_seg) != 0 && first_of_pair) {
    double2* o = reinterpret_cast<double2*>(A.chain_rec + (int64_t)row * CHAIN_REC + 6);
    o[0] = make_double2(HB[0], HB[1]);
    o[1] = make_double2(HB[2], HB[3]);
    o[2] = make_double2(HB[4], HB[5]);
    o[3] = make_double2(HB[6], HB[7]);
    A.chain_rec[(int64_t)row * CHAIN_REC + 14] = HB[8];
  }
  double d0 = S[0] * S[0] + S[3] * S[3] + S[6] * S[6], d1 = S[0] * S[1] + S[3] * S[4] + S[6] * S[7];
  double d2 = S[0] * S[2] + S[3] * S[5] + S[6] * S[8], d3 = S[1] * S[1] + S[4] * S[4] + S[7] * S[7];
  double d4 = S[1] * S[2] + S[4] * S[5] + S[7] * S[8], d5 = S[2] * S[2] + S[5] * S[5] + S[8] * S[8];
  double g0 = S[0] * R[R0] + S[3] * R[R0 + 1] + S[6] * R[R0 + 2], g1 = S[1] * R[R0] + S[4] * R[R0 + 1] + S[7] * R[R0 + 2];
  double g2 = S[2] * R[R0] + S[5] * R[R0 + 1] + S[8] * R[R0 + 2];
  if constexpr (SC) {
    scr[9][tid] = d0;  scr[10][tid] = d3; scr[11][tid] = d5;   // unreduced diagonal and gradient
    scr[12][tid] = g0; scr[13][tid] = g1; scr[14][tid] = g2;
    d0 -= cc * vs[0] * vs[0]; d1 -= cc * vs[0] * vs[1]; d2 -= cc * vs[0] * vs[2];
    d3 -= cc * vs[1] * vs[1]; d4 -= cc * vs[1] * vs[2]; d5 -= cc * vs[2] * vs[2];
    g0 -= gam * vs[0]; g1 -= gam * vs[1]; g2 -= gam * vs[2];  // J'(r - gamma j)
  }
  scr[0][tid] = d0; scr[1][tid] = d1; scr[2][tid] = d2; scr[3][tid] = d3; scr[4][tid] = d4; scr[5][tid] = d5;
  scr[6][tid] = g0; scr[7][tid] = g1; scr[8][tid] = g2;
}

template <bool SC>
__device__ __forceinline__ void asm_store_row(const AsmArgs& A, int c, int row, double s) {
  if (c < 6) {
    A.hd[(int64_t)c * A.n_loc + row] = s;
  } else if (c < 9) {
    A.gs[3 * (int64_t)row + (c - 6)] = s;
  } else if (c < 12) {
    A.diag_full[3 * (int64_t)row + (c - 9)] = s;
  } else {
    A.gs_full[3 * (int64_t)row + (c - 12)] = s;
  }
}

// One workgroup per tile (many workgroups: eight resident per compute unit hide the tile's one dependent round trip --
// index triple -> record / scales).  Round 3 took two dependent steps out of that chain: the tile's 16-byte descriptor
// replaces tile_row -> inc_ptr, and the row of an incidence comes from a u8 offset instead of an eight-step binary search
// in inc_ptr; the tile's row pointers wait in LDS for the row phase.  A persistent, software-pipelined form like K3's (index
// triples of the next tile prefetched, double-buffered staging, ~1000-1500 workgroups) was built and measured SLOWER:
// 505 us against 409 us at 1M poses -- with 39 KB of LDS only four workgroups fit a compute unit, and the records
// themselves (the long pole) were still requested only after the previous tile's barrier.
// (On the padded-slot layout, pgo::pad_tiles_to_slots, a variant that requests the tile's index triples at WG t without
// waiting for the descriptor was measured SLOWER: 424-430 against 413-415 us.)
template <bool SC, bool INFO>
__global__ __launch_bounds__(WG) void k_assemble(AsmArgs A) {
  static_assert(!(SC && INFO), "switchable constraints have no information-weighted form here");
  constexpr int NS = SC ? 15 : 9;  // staged values per incidence
  __shared__ double scr[NS][WG];
  __shared__ int sptr[WG + 1];
  const int tid = threadIdx.x;
  const XcdRange xr = xcd_range(A.n_tiles);
  for (int t = xr.begin; t < xr.end; t += xr.step) {
    const int4 d = A.tile_desc[t];
    const int r0 = d.x, nrows = d.y, q0 = d.z, nq = d.w;
    if (nq <= WG) {
      for (int k = tid; k <= nrows; k += WG) sptr[k] = A.inc_ptr[r0 + k] - q0;   // (nrows can be 256: rows without edges)
      if (tid < nq) {
        const int q = q0 + tid;
        const int ed = A.inc_edge[q], col = A.inc_col[q], roff = A.inc_rowoff[q];
        if (ed >= 0) {
          const bool first_of_pair = tid == 0 || roff != (int)A.inc_rowoff[q - 1] || A.inc_col[q - 1] != col;
          asm_incidence<SC, INFO>(A, q, ed, (int64_t)col, r0 + roff, first_of_pair, tid, scr);
        } else {   // a null incidence of a padded tile (pad_tiles_to_slots): contributes 0; its block stays the zero it was allocated as
#pragma unroll
          for (int c = 0; c < NS; ++c) scr[c][tid] = 0.0;
        }
      }
      __syncthreads();
      for (int idx = tid; idx < nrows * NS; idx += WG) {
        const int c = idx / nrows, rl = idx - c * nrows;
        const int lo = sptr[rl], hi = sptr[rl + 1];
        double s = 0.0;
        for (int j = lo; j < hi; ++j) s += scr[c][j];
        asm_store_row<SC>(A, c, r0 + rl, s);
      }
      __syncthreads();
    } else {
      // ---- one heavy row (> 256 incidences): chunks of 256, two barriers per chunk, the sums carried in registers
      const int q1 = q0 + nq;
      double acc = 0.0;  // tid < NS
      for (int base = q0; base < q1; base += WG) {
        const int q = base + tid;
        if (q < q1) {
          const int e2 = A.inc_edge[q];
          const int c2 = A.inc_col[q];
          const bool first_of_pair = q == q0 || A.inc_col[q - 1] != c2;
          asm_incidence<SC, INFO>(A, q, e2, (int64_t)c2, r0, first_of_pair, tid, scr);
        }
        __syncthreads();
        if (tid < NS) {
          const int hi = min(q1, base + WG) - base;
          double s = 0.0;
          for (int j = 0; j < hi; ++j) s += scr[tid][j];
          acc += s;
        }
        __syncthreads();
      }
      if (tid < NS) asm_store_row<SC>(A, tid, r0, acc);
    }
  }
}

// ------------------------------------------------------------------- K3
struct CgState {       // lives in device memory
  double rz[2];        // r.z, double-buffered by iteration parity
  double bb;           // b.b
  double rr;           // r.r after the latest update
  double tol2;         // (rtol^2) b.b
  int32_t done;
  int32_t iters;
  int32_t pending;     // fused-update loop (k_spmv MODE 5): an iteration's r.z / r.r partials wait to be booked
  int32_t started;     // fused-update loop: 0 until the first k_cg_update1 of the solve has run.  Written by k_cg_init_fin and
                       // k_cg_update1* only -- never during a k_spmv launch, whose workgroups all read it
  // single-reduction loop (k_cg_sr_*): coefficients of the coming vector update, written by k_cg_sr_scal only
  double sr_alpha, sr_beta, sr_gamma;
};

struct SpmvArgs {
  const int32_t* inc_ptr;
  const int32_t* inc_col;    // global pose position of the column block
  const int32_t* tile_row;
  const int4* tile_desc;     // per tile {first local row, rows, first incidence, incidences}
  int32_t n_tiles;
  int32_t n_loc;
  int32_t lo;                // first owned global row
  int32_t with_d2;
  int32_t nt;                // non-temporal loads of the H stream
  int64_t inc_stride;
  const double* hoff;        // AoSoA, see hoff_index
  const double* hd;          // 6 planes
  const double* d2;          // [n_loc x 3] LM diagonal D'D
  const double* hdd;         // 3 planes [n_loc]: diagonal of (J_self)'(J_self) + D'D (k_prepare); read by k_spmv_p instead of
                             // the three diagonal planes of hd AND d2 -- 24 B per row less
  const double* p;           // gathered vector, GLOBAL indexing [.. x 3]
  double* y;                 // [n_loc x 3]
  double* dot_part;          // [gridDim.x] partial of p_owned . y
  const int32_t* done;       // skip when *done != 0 (nullptr: never)
  // MODE 5 only (small graphs, one rank): the direction update p = z + beta p of the PREVIOUS iteration is applied on
  // the fly -- gathered as z[col] + beta p_old[col], written for the tile's own rows to p_new -- so that a PCG iteration
  // is two launches instead of three.  A.p is p_old; beta and the convergence decision come from the r.z / r.r partials
  // of the previous k_cg_update1, re-summed by every workgroup; workgroup 0 books them in st (once: st->pending).
  const double* z;           // [n_loc x 3]
  double* p_new;             // gather vector of this iteration (the other buffer of the pair)
  const double* part_rz;
  const double* part_rr;
  int32_t n_rz, n_rr, parity;  // parity of the iteration whose partials these are
  CgState* st;
};

// Algorithmic bytes: 76 per off-diagonal block (72 value + 4 column index) + per row
// 48 (diagonal planes) + 24 (D'D) + 4 (row pointer) + 24 (y) + 24 (p, counted once).
// MODE 0 = the product kernel.  MODE 1..3 are timing-only ablations used by pgo_bench_spmv under
// PGO_SPMV_ABLATE (1: no p[col] gather, 2: no H-plane loads, 3: neither): wrong results, same structure.
// MODE 4 = the part of the product that needs OWNED columns only (blocks whose column lies on another rank contribute
// 0 and are not loaded): it runs while the halo exchange is in flight, k_spmv_remote adds the rest afterwards.
template <int MODE>
__global__ __launch_bounds__(WG) void k_spmv_t(SpmvArgs A) {
  // double-buffered staging: ONE barrier per tile (the barrier of tile t+1 orders every wave's row
  // phase of tile t before any wave's lane phase of tile t+2, which reuses the buffer)
  __shared__ double scr[2][3][WG];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  if (A.done && *A.done) return;
  double beta = 0.0;
  // gathered search direction of pose `col` (global index)
  auto load_p = [&](int64_t col, double& p0, double& p1, double& p2) {
    gather3(A.p, col, p0, p1, p2);
    if (MODE == 5) {
      const double* zc = A.z + 3 * (col - A.lo);
      p0 = zc[0] + beta * p0;
      p1 = zc[1] + beta * p1;
      p2 = zc[2] + beta * p2;
    }
  };
  double dot = 0.0;
  int buf = 0;
  const int64_t n = A.n_loc;
  const XcdRange xr = xcd_range(A.n_tiles);
  // A tile used to cost four DEPENDENT memory round trips (tile_row -> inc_ptr -> inc_col -> p[col]); with eight
  // workgroups per CU already resident that chain, not bandwidth, set the pace (structure alone 44 us, + H stream
  // 150 us, + gather 190 us at 1M poses).  Now the tile's descriptor {first row, rows, first incidence, incidences} is one
  // 16-byte record, and the NEXT tile's descriptor and column indices are fetched while the current tile is in flight:
  // per tile one round trip (H stream + gather + row operands, all independent) remains.
  int t = xr.begin;
  int4 d = make_int4(0, 0, 0, 0);
  int col_pf = 0;
  if (t < xr.end) {
    d = A.tile_desc[t];
    if (d.w <= WG && tid < d.w) col_pf = ld_stream(A.inc_col + d.z + tid);
  }
  if (MODE == 5) {  // (after the first tile's loads are on their way: independent of them)
    const double rz_new = sum_partials_bcast(A.part_rz, A.n_rz, red);
    const double rr = sum_partials_bcast(A.part_rr, A.n_rr, red);
    const double rz_old = A.st->rz[A.parity], tol2 = A.st->tol2;
    if (blockIdx.x == 0 && tid == 0 && A.st->pending) {  // book the previous iteration (k_cg_update2's scalar part)
      A.st->rz[A.parity ^ 1] = rz_new;
      A.st->rr = rr;
      A.st->iters += 1;
      A.st->pending = 0;
      if (rr <= tol2) A.st->done = 1;
    }
    // converged (same decision everywhere; never in the first product of a solve, whose partials are stale).  `started`
    // is not written during this launch -- iters / pending are, by workgroup 0 above
    if (rr <= tol2 && A.st->started) return;
    beta = rz_new / rz_old;
  }
  for (; t < xr.end; t += xr.step) {
    const int r0 = d.x, nrows = d.y, q0 = d.z, nq = d.w;
    const int q1 = q0 + nq;
    const int tn = t + xr.step;
    int4 dn = d;
    if (tn < xr.end) dn = A.tile_desc[tn];   // oldest load of the iteration: back long before it is needed
    int col_n = 0;
    if (nq <= WG) {
      // ---- common case: the tile is one chunk.  Row-phase operands are fetched up front so that
      // their latency overlaps the lane phase instead of following the barrier.
      const bool pv = tid < nrows * 3;
      int a = 0, row = r0, lo = 0, hi = 0;
      double h0 = 0.0, h1 = 0.0, h2 = 0.0, dd = 0.0, pr0 = 0.0, pr1 = 0.0, pr2 = 0.0;
      if (pv) {
        a = tid / nrows;
        row = r0 + (tid - a * nrows);
        lo = A.inc_ptr[row] - q0;
        hi = A.inc_ptr[row + 1] - q0;
        const int i1 = (a == 0) ? 1 : (a == 1 ? 3 : 4), i2 = (a == 2) ? 5 : (a == 1 ? 4 : 2);
        h0 = ld_stream(A.hd + ((int64_t)a * n + row));
        h1 = ld_stream(A.hd + ((int64_t)i1 * n + row));
        h2 = ld_stream(A.hd + ((int64_t)i2 * n + row));
        if (A.with_d2) dd = ld_stream(A.d2 + (3 * (int64_t)row + a));
        load_p((int64_t)A.lo + row, pr0, pr1, pr2);
        if (MODE == 5) A.p_new[PS * (int64_t)(A.lo + row) + a] = (a == 0) ? pr0 : (a == 1 ? pr1 : pr2);
      }
      const int q = q0 + tid;
      const bool lane_on = q < q1;
      const int64_t col = col_pf;
      const bool skip = MODE == 4 && (col < A.lo || col >= (int64_t)A.lo + A.n_loc);
      double p0 = 0.0, p1 = 0.0, p2 = 0.0, h[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (lane_on && !skip) {
        if (MODE == 1 || MODE == 3) {
          p0 = (double)col; p1 = p0 + 1.0; p2 = p0 + 2.0;
        } else {
          load_p(col, p0, p1, p2);
        }
        if (MODE != 2 && MODE != 3) {
          if (A.nt) hoff_load_nt(A.hoff, q, h);
          else hoff_load(A.hoff, q, h);
        }
      }
      // the next tile's column indices (its descriptor has arrived: it was the first load of this iteration)
      if (tn < xr.end && dn.w <= WG && tid < dn.w) col_n = ld_stream(A.inc_col + dn.z + tid);
      if (lane_on) {
        if (MODE == 2 || MODE == 3) {
          scr[buf][0][tid] = p0 + 2.0 * p1 + 3.0 * p2;
          scr[buf][1][tid] = p0 - p1;
          scr[buf][2][tid] = p2 * p1;
        } else {
          scr[buf][0][tid] = h[0] * p0 + h[1] * p1 + h[2] * p2;   // skipped (remote-column) blocks: h = 0
          scr[buf][1][tid] = h[3] * p0 + h[4] * p1 + h[5] * p2;
          scr[buf][2][tid] = h[6] * p0 + h[7] * p1 + h[8] * p2;
        }
      }
      __syncthreads();
      if (pv) {
        double s = 0.0;
        for (int j = lo; j < hi; ++j) s += scr[buf][a][j];
        const double pa = (a == 0) ? pr0 : (a == 1 ? pr1 : pr2);
        s += h0 * pr0 + h1 * pr1 + h2 * pr2 + dd * pa;
        st_stream(A.y + (3 * (int64_t)row + a), s);
        dot += pa * s;
      }
      for (int idx = tid + WG; idx < nrows * 3; idx += WG) {  // tiles of very low-degree rows (> 85 rows)
        const int a2 = idx / nrows, row2 = r0 + (idx - a2 * nrows);
        const int lo2 = A.inc_ptr[row2] - q0, hi2 = A.inc_ptr[row2 + 1] - q0;
        double s = 0.0;
        for (int j = lo2; j < hi2; ++j) s += scr[buf][a2][j];
        double pr[3];
        load_p((int64_t)A.lo + row2, pr[0], pr[1], pr[2]);
        if (MODE == 5) A.p_new[PS * (int64_t)(A.lo + row2) + a2] = pr[a2];
        const int i1 = (a2 == 0) ? 1 : (a2 == 1 ? 3 : 4), i2 = (a2 == 2) ? 5 : (a2 == 1 ? 4 : 2);
        double dg = A.hd[(int64_t)a2 * n + row2] * pr[0];
        dg += A.hd[(int64_t)i1 * n + row2] * pr[1];
        dg += A.hd[(int64_t)i2 * n + row2] * pr[2];
        if (A.with_d2) dg += A.d2[3 * (int64_t)row2 + a2] * pr[a2];
        s += dg;
        A.y[3 * (int64_t)row2 + a2] = s;
        dot += pr[a2] * s;
      }
      buf ^= 1;
    } else {
      // ---- one heavy row (> 256 incidences): chunked, two barriers per chunk
      double acc = 0.0;
      for (int base = q0; base < q1; base += WG) {
        const int q = base + tid;
        if (q < q1) {
          const int64_t col = A.inc_col[q];
          if (MODE == 4 && (col < A.lo || col >= (int64_t)A.lo + A.n_loc)) {
            scr[buf][0][tid] = 0.0;
            scr[buf][1][tid] = 0.0;
            scr[buf][2][tid] = 0.0;
          } else {
            double p0, p1, p2, h[9];
            load_p(col, p0, p1, p2);
            hoff_load(A.hoff, q, h);
            scr[buf][0][tid] = h[0] * p0 + h[1] * p1 + h[2] * p2;
            scr[buf][1][tid] = h[3] * p0 + h[4] * p1 + h[5] * p2;
            scr[buf][2][tid] = h[6] * p0 + h[7] * p1 + h[8] * p2;
          }
        }
        __syncthreads();
        if (tid < 3) {
          const int hi = min(q1, base + WG) - base;
          double s = 0.0;
          for (int j = 0; j < hi; ++j) s += scr[buf][tid][j];
          acc += s;
        }
        __syncthreads();
      }
      if (tid < 3) {
        const int a = tid, row = r0;
        double pr[3];
        load_p((int64_t)A.lo + row, pr[0], pr[1], pr[2]);
        if (MODE == 5) A.p_new[PS * (int64_t)(A.lo + row) + a] = pr[a];
        const int i1 = (a == 0) ? 1 : (a == 1 ? 3 : 4), i2 = (a == 2) ? 5 : (a == 1 ? 4 : 2);
        double dg = A.hd[(int64_t)a * n + row] * pr[0];
        dg += A.hd[(int64_t)i1 * n + row] * pr[1];
        dg += A.hd[(int64_t)i2 * n + row] * pr[2];
        if (A.with_d2) dg += A.d2[3 * (int64_t)row + a] * pr[a];
        const double s = acc + dg;
        A.y[3 * (int64_t)row + a] = s;
        dot += pr[a] * s;
      }
      if (tn < xr.end && dn.w <= WG && tid < dn.w) col_n = A.inc_col[dn.z + tid];
    }
    d = dn;
    col_pf = col_n;
  }
  const double tot = block_sum_bcast(dot, red);
  if (tid == 0) A.dot_part[blockIdx.x] = tot;
}

// ------------------------------------------------------------------- K3, software-pipelined form
// The same product (single-rank product kernel: MODE 0 of k_spmv_t, tiles of <= 256 incidences only) with the loads of
// tile t + 1 -- block stream, gathers, row operands -- issued BEFORE the barrier and the row phase of tile t, so that a
// workgroup always has a tile's worth of memory requests in flight instead of waiting out one round trip per tile.
// Costs registers (two tiles' operands live: ~110 VGPRs, 4 workgroups per CU instead of 8) -- the same bytes in flight.
// PSTR: doubles per pose of the gathered vector.  PS in the product; experiment builds (scripts/exp_mall.sh) time the kernel
// on a copy of p spread over 96 / 288 bytes per pose -- a table of 96 / 288 MB at 1M poses, i.e. inside / beyond the
// 256 MiB Infinity Cache -- to tell gathers served by that cache from gathers served by HBM.
template <int PSTR = PS>
__global__ __launch_bounds__(WG) void k_spmv_p(SpmvArgs A) {
  __shared__ double scr[2][3][WG];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  if (A.done && *A.done) return;
  const int64_t n = A.n_loc;
  const XcdRange xr = xcd_range(A.n_tiles);
  struct Lane {
    double h[9], p0, p1, p2;     // lane phase: block + gathered direction
    double h0, h1, h2, pr0, pr1, pr2;       // row phase: row a of the diagonal block (LM diagonal folded in), own direction
    int a, row, lo, hi;
    bool on, pv;
  };
  auto issue = [&](const int4& d, int col_) {
    Lane L;
    const int r0 = d.x, nrows = d.y, q0 = d.z, nq = d.w;
    L.pv = tid < nrows * 3;
    L.a = 0; L.row = r0; L.lo = 0; L.hi = 0;
    L.h0 = L.h1 = L.h2 = L.pr0 = L.pr1 = L.pr2 = 0.0;
    if (L.pv) {
      L.a = tid / nrows;
      L.row = r0 + (tid - L.a * nrows);
      L.lo = A.inc_ptr[L.row] - q0;
      L.hi = A.inc_ptr[L.row + 1] - q0;
      const int a = L.a;
      // symmetric row a of the diagonal block: (h0, h1, h2) multiply (p0, p1, p2); the diagonal entry comes with D'D
      // already added (hdd) when the product includes the LM diagonal
      const int i0 = (a == 0) ? 0 : a, i1 = (a == 0) ? 1 : (a == 1 ? 3 : 4), i2 = (a == 2) ? 5 : (a == 1 ? 4 : 2);
      const double* hdg = A.with_d2 ? A.hdd + ((int64_t)a * n + L.row) : A.hd + ((int64_t)(a == 0 ? 0 : (a == 1 ? 3 : 5)) * n + L.row);
      const double dg = ld_stream(hdg);
      const double o0 = (a == 0) ? 0.0 : ld_stream(A.hd + ((int64_t)i0 * n + L.row));
      const double o1 = (a == 1) ? 0.0 : ld_stream(A.hd + ((int64_t)i1 * n + L.row));
      const double o2 = (a == 2) ? 0.0 : ld_stream(A.hd + ((int64_t)i2 * n + L.row));
      L.h0 = (a == 0) ? dg : o0;
      L.h1 = (a == 1) ? dg : o1;
      L.h2 = (a == 2) ? dg : o2;
      gather3<PSTR>(A.p, (int64_t)A.lo + L.row, L.pr0, L.pr1, L.pr2);
    }
    L.on = tid < nq;
    L.p0 = L.p1 = L.p2 = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) L.h[c] = 0.0;
    if (L.on) {
      gather3<PSTR>(A.p, (int64_t)col_, L.p0, L.p1, L.p2);
      if (A.nt) hoff_load_nt(A.hoff, q0 + tid, L.h);
      else hoff_load(A.hoff, q0 + tid, L.h);
    }
    return L;
  };
  double dot = 0.0;
  int buf = 0;
  int t = xr.begin;
  if (t < xr.end) {
    int4 d0 = A.tile_desc[t];
    int col0 = (tid < d0.w) ? A.inc_col[d0.z + tid] : 0;
    Lane L = issue(d0, col0);
    // descriptor and column indices of the tile after: one iteration ahead of their use
    int4 d1 = d0;
    int col1 = 0;
    if (t + xr.step < xr.end) {
      d1 = A.tile_desc[t + xr.step];
      col1 = (tid < d1.w) ? ld_stream(A.inc_col + d1.z + tid) : 0;
    }
    for (; t < xr.end; t += xr.step) {
      const int tn = t + xr.step, tnn = tn + xr.step;
      int4 d2 = d1;
      if (tnn < xr.end) d2 = A.tile_desc[tnn];
      // stage tile t
      if (L.on) {
        scr[buf][0][tid] = L.h[0] * L.p0 + L.h[1] * L.p1 + L.h[2] * L.p2;
        scr[buf][1][tid] = L.h[3] * L.p0 + L.h[4] * L.p1 + L.h[5] * L.p2;
        scr[buf][2][tid] = L.h[6] * L.p0 + L.h[7] * L.p1 + L.h[8] * L.p2;
      }
      // row operands of tile t stay; everything of tile t + 1 goes out now
      const bool pv = L.pv;
      const int a = L.a, row = L.row, lo = L.lo, hi = L.hi;
      const double h0 = L.h0, h1 = L.h1, h2 = L.h2, pr0 = L.pr0, pr1 = L.pr1, pr2 = L.pr2;
      if (tn < xr.end) L = issue(d1, col1);
      int col2 = 0;
      if (tnn < xr.end) col2 = (tid < d2.w) ? ld_stream(A.inc_col + d2.z + tid) : 0;
      __syncthreads();
      if (pv) {
        double s = 0.0;
        for (int j = lo; j < hi; ++j) s += scr[buf][a][j];
        const double pa = (a == 0) ? pr0 : (a == 1 ? pr1 : pr2);
        s += h0 * pr0 + h1 * pr1 + h2 * pr2;
        st_stream(A.y + (3 * (int64_t)row + a), s);
        dot += pa * s;
      }
      buf ^= 1;
      d1 = d2;
      col1 = col2;
    }
  }
  const double tot = block_sum_bcast(dot, red);
  if (tid == 0) A.dot_part[blockIdx.x] = tot;
}

// ------------------------------------------------------------------- K3, one tile per workgroup
// The same product once more, in the opposite style: NO software pipeline, one tile per workgroup and as many workgroups
// as tiles -- 31.8k at 1M poses -- so that the hardware's workgroup scheduler does the overlapping: short workgroups, each
// one dependent chain (descriptor -> column index -> block + gather -> row phase), a new one starting whenever one
// retires.  Measured on the box where the pipelined k_spmv_p (1024 persistent workgroups) takes 164-166 us: k_spmv_t with
// 2048 / 4096 / 8192 / 16384 / 31808 workgroups 188 / 179 / 173 / 167 / 154 us, this kernel 151 us.  Plain tiles only
// (<= TW incidences, <= TW / 3 rows, TW threads); its dot partials -- one per tile -- are folded by k_fold_partials.
// PAD: the padded-slot layout (pgo::pad_tiles_to_slots): tile t's column indices and blocks sit at WG t, every lane has a slot
// (null incidences: zero block), so they are requested WITHOUT waiting for the tile's descriptor -- one dependent round trip
// less per workgroup.
// (Measured and not kept: finer tiles of K3's own, 128 / 64 incidences and threads per workgroup -- 140-142 / 159-166 us against
// 144-145; 2 / 4 tiles per workgroup of 512 / 1024 threads, 2 / 4 times fewer partials -- 41.3-41.5 / 39.7-40.1 against 42.0 GN it/s.)
template <bool PAD = false>
__global__ __launch_bounds__(WG) void k_spmv_1(SpmvArgs A) {
  __shared__ double scr[3][WG];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  if (A.done && *A.done) return;
  const int64_t n = A.n_loc;
  const XcdRange xr = xcd_range(A.n_tiles);
  double dot = 0.0;
  const int t = xr.begin;
  if (xr.begin < xr.end) {       // (as many workgroups as tiles: at most one per workgroup)
    const int4 d = A.tile_desc[t];
    const int r0 = d.x, nrows = d.y, nq = d.w;
    const int q0 = PAD ? WG * t : d.z;
    const bool on = PAD ? true : tid < nq, pv = tid < nrows * 3;
    int col = 0;
    if (on) col = ld_stream(A.inc_col + q0 + tid);
    // row operands first (independent of the column index): row a of the diagonal block with D'D folded in, own direction
    int a = 0, row = r0, lo = 0, hi = 0;
    double h0 = 0.0, h1 = 0.0, h2 = 0.0, pr0 = 0.0, pr1 = 0.0, pr2 = 0.0;
    if (pv) {
      a = tid / nrows;
      row = r0 + (tid - a * nrows);
      lo = A.inc_ptr[row] - d.z;
      hi = A.inc_ptr[row + 1] - d.z;
      const int i0 = (a == 0) ? 0 : a, i1 = (a == 0) ? 1 : (a == 1 ? 3 : 4), i2 = (a == 2) ? 5 : (a == 1 ? 4 : 2);
      const double* hdg = A.with_d2 ? A.hdd + ((int64_t)a * n + row) : A.hd + ((int64_t)(a == 0 ? 0 : (a == 1 ? 3 : 5)) * n + row);
      const double dg = ld_stream(hdg);
      const double o0 = (a == 0) ? 0.0 : ld_stream(A.hd + ((int64_t)i0 * n + row));
      const double o1 = (a == 1) ? 0.0 : ld_stream(A.hd + ((int64_t)i1 * n + row));
      const double o2 = (a == 2) ? 0.0 : ld_stream(A.hd + ((int64_t)i2 * n + row));
      h0 = (a == 0) ? dg : o0;
      h1 = (a == 1) ? dg : o1;
      h2 = (a == 2) ? dg : o2;
      gather3(A.p, (int64_t)A.lo + row, pr0, pr1, pr2);
    }
    if (on) {
      double h[9], p0, p1, p2;
      if (A.nt) hoff_load_nt(A.hoff, q0 + tid, h);
      else hoff_load(A.hoff, q0 + tid, h);
      gather3(A.p, (int64_t)col, p0, p1, p2);
      scr[0][tid] = h[0] * p0 + h[1] * p1 + h[2] * p2;
      scr[1][tid] = h[3] * p0 + h[4] * p1 + h[5] * p2;
      scr[2][tid] = h[6] * p0 + h[7] * p1 + h[8] * p2;
    }
    __syncthreads();
    if (pv) {
      double s = 0.0;
      for (int j = lo; j < hi; ++j) s += scr[a][j];
      const double pa = (a == 0) ? pr0 : (a == 1 ? pr1 : pr2);
      s += h0 * pr0 + h1 * pr1 + h2 * pr2;
      st_stream(A.y + (3 * (int64_t)row + a), s);
      dot += pa * s;
    }
  }
  dot = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  if (tid == 0) A.dot_part[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// The blocks whose column lives on another rank (a few % of a shard's incidences, listed per row at create): after the
// halo exchange, y_row += sum H_rc p_c and the matching part of p . A p.  One thread per row that has such blocks.
struct RemoteArgs {
  const int32_t* rows;     // local rows with remote-column blocks
  const int32_t* ptr;      // n_rows + 1 into slots
  const int32_t* slots;    // incidence positions q (the block's place in hoff / inc_col)
  const int32_t* inc_col;
  const double* hoff;
  const double* p;         // gathered vector, global indexing
  double* y;
  double* dot_part;
  int32_t n_rows;
  int32_t lo;
  const int32_t* done;
};
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_spmv_remote(RemoteArgs A) {
  __shared__ double red[8];
  if (A.done && *A.done) return;
  double dot = 0.0;
  for (int i = blockIdx.x * WG + threadIdx.x; i < A.n_rows; i += gridDim.x * WG) {
    const int row = A.rows[i];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int q = A.slots[k];
      const int64_t col = A.inc_col[q];
      double p0, p1, p2, h[9];
      gather3(A.p, col, p0, p1, p2);
      hoff_load(A.hoff, q, h);
      s0 += h[0] * p0 + h[1] * p1 + h[2] * p2;
      s1 += h[3] * p0 + h[4] * p1 + h[5] * p2;
      s2 += h[6] * p0 + h[7] * p1 + h[8] * p2;
    }
    double* y = A.y + 3 * (int64_t)row;
    const double* pr = A.p + PS * (int64_t)(A.lo + row);
    y[0] += s0;
    y[1] += s1;
    y[2] += s2;
    dot += pr[0] * s0 + pr[1] * s1 + pr[2] * s2;
  }
  const double tot = block_sum_bcast(dot, red);
  if (threadIdx.x == 0) A.dot_part[blockIdx.x] = tot;
}

// ------------------------------------------------------ per-row kernels
// symmetric plane indices: 0:d00 1:d01 2:d02 3:d11 4:d12 5:d22

// Jacobi column scaling 1/(1 + ||J col||) from the unscaled diagonal (Ceres
// TrustRegionMinimizer, iteration 0); 0 on the constant pose.
// fixed_mask (batched handles: one anchored pose per problem + the padding rows): nullptr = only pose `fixed` is constant
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_jacobi_scale(const double* __restrict__ hd, int n_loc, int lo, int fixed, int enabled,
                               double* __restrict__ scale, const uint8_t* __restrict__ fixed_mask) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_loc) return;
  if (fixed_mask && fixed_mask[row]) fixed = lo + row;
  const int64_t n = n_loc;
  double s0 = 1.0, s1 = 1.0, s2 = 1.0;
  if (enabled) {
    s0 = 1.0 / (1.0 + sqrt(hd[row]));
    s1 = 1.0 / (1.0 + sqrt(hd[3 * n + row]));
    s2 = 1.0 / (1.0 + sqrt(hd[5 * n + row]));
  }
  if (lo + row == fixed) s0 = s1 = s2 = 0.0;
  double* o = scale + 3 * (int64_t)(lo + row);
  o[0] = s0;
  o[1] = s1;
  o[2] = s2;
}

// LM diagonal D'D = clamp(diag(H), min, max) / radius and the block-Jacobi
// preconditioner M^-1 = (H_ii + D'D)^-1  (symmetric 3x3, 6 planes).
// diag_full (METHOD 2): the UNREDUCED squared column norms the LM diagonal is defined on; nullptr = hd's diagonal
// Batched handles: the trust-region radius is per problem -- prob_radius[prob_of_256[row >> 8]] (problems start at
// multiples of 256 rows) -- and fixed_mask marks each problem's anchored pose and the padding rows.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_prepare(const double* __restrict__ hd, const double* __restrict__ diag_full, int n_loc, int lo, int fixed,
                          double radius, double dmin, double dmax, double* __restrict__ d2, double* __restrict__ minv,
                          const uint8_t* __restrict__ fixed_mask, const int32_t* __restrict__ prob_of_256,
                          const double* __restrict__ prob_radius, double* __restrict__ chain_rec, double* __restrict__ hdd) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_loc) return;
  if (fixed_mask && fixed_mask[row]) fixed = lo + row;
  if (prob_radius) radius = prob_radius[prob_of_256[row >> 8]];
  const int64_t n = n_loc;
  double a00 = hd[row], a01 = hd[n + row], a02 = hd[2 * n + row], a11 = hd[3 * n + row], a12 = hd[4 * n + row],
         a22 = hd[5 * n + row];
  const double n0 = diag_full ? diag_full[3 * (int64_t)row] : a00, n1 = diag_full ? diag_full[3 * (int64_t)row + 1] : a11,
               n2 = diag_full ? diag_full[3 * (int64_t)row + 2] : a22;
  double e0 = fmin(fmax(n0, dmin), dmax) / radius;
  double e1 = fmin(fmax(n1, dmin), dmax) / radius;
  double e2 = fmin(fmax(n2, dmin), dmax) / radius;
  if (lo + row == fixed) e0 = e1 = e2 = 1.0;  // decoupled identity row: the constant pose never moves
  d2[3 * (int64_t)row] = e0;
  d2[3 * (int64_t)row + 1] = e1;
  d2[3 * (int64_t)row + 2] = e2;
  a00 += e0;
  a11 += e1;
  a22 += e2;
  if (hdd) {   // diagonal of H + D'D for the product kernel (k_spmv_p)
    hdd[row] = a00;
    hdd[n + row] = a11;
    hdd[2 * n + row] = a22;
  }
  if (chain_rec) {  // chain preconditioner: M_ii = H_ii + D'D into the factorisation's input record (the 3x3 inverse below is unused)
    double2* o = reinterpret_cast<double2*>(chain_rec + (int64_t)row * CHAIN_REC);
    o[0] = make_double2(a00, a01);
    o[1] = make_double2(a02, a11);
    o[2] = make_double2(a12, a22);
    return;
  }
  const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double id = 1.0 / det;
  minv[row] = c00 * id;
  minv[n + row] = c01 * id;
  minv[2 * n + row] = c02 * id;
  minv[3 * n + row] = (a00 * a22 - a02 * a02) * id;
  minv[4 * n + row] = (a01 * a02 - a00 * a12) * id;
  minv[5 * n + row] = (a00 * a11 - a01 * a01) * id;
}

// max_i |g_i| of the UNSCALED gradient g = gs / s over the free parameters (Ceres
// gradient_max_norm); partial max per workgroup.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_grad_max(const double* __restrict__ gs, const double* __restrict__ scale,
                                                 int n_loc, int lo, double* __restrict__ part) {
  __shared__ double red[8];
  double m = 0.0;
  const int64_t n3 = 3 * (int64_t)n_loc;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n3; i += (int64_t)gridDim.x * WG) {
    const double s = scale[3 * (int64_t)lo + i];
    if (s > 0.0) m = fmax(m, fabs(gs[i] / s));
  }
  m = block_max_bcast(m, red);
  if (threadIdx.x == 0) part[blockIdx.x] = m;
}

// out[k] = reduce(part_k[0..n_k)) for up to 4 partial arrays; one workgroup.
struct FinArgs {
  const double* part[6];
  int32_t n[6];
  int32_t is_max[6];
  int32_t count;
  double* out;
};
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_finalize(FinArgs A) {
  __shared__ double red[8];
  for (int k = 0; k < A.count; ++k) {
    double v = 0.0;
    if (A.is_max[k]) {
      for (int i = threadIdx.x; i < A.n[k]; i += WG) v = fmax(v, A.part[k][i]);
      v = block_max_bcast(v, red);
    } else {
      v = sum_partials_bcast(A.part[k], A.n[k], red);
    }
    if (threadIdx.x == 0) A.out[k] = v;
  }
}

// n partials -> gridDim.x partials (fixed order): workgroup b sums its contiguous share, every thread a handful of elements
// requested together.  Between k_spmv_1 (one dot partial per tile: 31.8k at 1M poses) and the update kernel, whose workgroups
// each re-sum the partials they are given: one workgroup summing 31.8k values alone took ~15 us of every PCG iteration.
template <int PGO_UNIT_ = 0>
__global__ __launch_bounds__(WG) void k_fold_partials(const double* __restrict__ in, int n, double* __restrict__ out,
                                                      const int32_t* __restrict__ done) {
  __shared__ double red[8];
  if (done && *done) return;
  const int chunk = (n + gridDim.x - 1) / gridDim.x;
  const int b0 = blockIdx.x * chunk, b1 = min(n, b0 + chunk);
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = b0 + threadIdx.x + k * WG;
    v[k] = i < b1 ? in[i] : 0.0;
  }
  double s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  for (int i = b0 + threadIdx.x + 8 * WG; i < b1; i += WG) s += in[i];
  s = block_sum_bcast(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ------------------------------------------------------------------- K5

struct CgVec {
  int32_t n_loc;
  int32_t lo;
  const double* minv;  // 6 planes
  double* y;           // solution   [n_loc x 3]
  double* r;           // residual
  double* z;           // M^-1 r
  double* ap;          // A p
  double* p;           // search direction, GLOBAL indexing (gathered by K3)
  CgState* st;
  int32_t fused;       // 1: the direction update is fused into the next k_spmv (MODE 5); update1 then flags its partials pending
  int32_t _pad;
};

__device__ __forceinline__ void minv_apply(const double* __restrict__ minv, int64_t n, int row, double r0, double r1,
                                           double r2, double& z0, double& z1, double& z2) {
  const double m00 = minv[row], m01 = minv[n + row], m02 = minv[2 * n + row], m11 = minv[3 * n + row],
               m12 = minv[4 * n + row], m22 = minv[5 * n + row];
  z0 = m00 * r0 + m01 * r1 + m02 * r2;
  z1 = m01 * r0 + m11 * r1 + m12 * r2;
  z2 = m02 * r0 + m12 * r1 + m22 * r2;
}

// y = 0, r = b, z = M^-1 r, p = z; partials of r.z and b.b
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_init(CgVec V, const double* __restrict__ b, double* __restrict__ part_rz,
                                                double* __restrict__ part_bb) {
  __shared__ double red[8];
  double rz = 0.0, bb = 0.0;
  const int64_t n = V.n_loc;
  for (int row = blockIdx.x * WG + threadIdx.x; row < V.n_loc; row += gridDim.x * WG) {
    const double r0 = b[3 * (int64_t)row], r1 = b[3 * (int64_t)row + 1], r2 = b[3 * (int64_t)row + 2];
    double z0, z1, z2;
    minv_apply(V.minv, n, row, r0, r1, r2, z0, z1, z2);
    double* o;
    o = V.y + 3 * (int64_t)row; o[0] = 0.0; o[1] = 0.0; o[2] = 0.0;
    o = V.r + 3 * (int64_t)row; o[0] = r0; o[1] = r1; o[2] = r2;
    o = V.z + 3 * (int64_t)row; o[0] = z0; o[1] = z1; o[2] = z2;
    o = V.p + PS * (int64_t)(V.lo + row); o[0] = z0; o[1] = z1; o[2] = z2;
    rz += r0 * z0 + r1 * z1 + r2 * z2;
    bb += r0 * r0 + r1 * r1 + r2 * r2;
  }
  rz = block_sum_bcast(rz, red);
  bb = block_sum_bcast(bb, red);
  if (threadIdx.x == 0) {
    part_rz[blockIdx.x] = rz;
    part_bb[blockIdx.x] = bb;
  }
}

// scal[0] = r.z, scal[1] = b.b (already reduced over workgroups and ranks)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_cg_init_fin(CgState* st, const double* __restrict__ scal, double rtol) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->rz[0] = scal[0];
    st->rz[1] = scal[0];   // (finite beta for the fused loop's first, no-op direction update)
    st->bb = scal[1];
    st->rr = scal[1];
    st->tol2 = rtol * rtol * scal[1];
    st->done = (scal[1] == 0.0) ? 1 : 0;
    st->iters = 0;
    st->pending = 0;
    st->started = 0;
  }
}

// alpha = rz / p.Ap ; y += alpha p ; r -= alpha Ap ; z = M^-1 r ; partials r.z, r.r
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_update1(CgVec V, int parity, const double* __restrict__ part_pap, int n_pap,
                                                   double* __restrict__ part_rz, double* __restrict__ part_rr) {
  __shared__ double red[8];
  if (V.st->done) return;
  const double pap = sum_partials_bcast(part_pap, n_pap, red);
  const double alpha = V.st->rz[parity] / pap;
  double rz = 0.0, rr = 0.0;
  const int64_t n = V.n_loc;
  for (int row = blockIdx.x * WG + threadIdx.x; row < V.n_loc; row += gridDim.x * WG) {
    const double* pp = V.p + PS * (int64_t)(V.lo + row);
    double* yy = V.y + 3 * (int64_t)row;
    double* rp = V.r + 3 * (int64_t)row;
    const double* ap = V.ap + 3 * (int64_t)row;
    yy[0] += alpha * pp[0];
    yy[1] += alpha * pp[1];
    yy[2] += alpha * pp[2];
    const double r0 = rp[0] - alpha * ap[0], r1 = rp[1] - alpha * ap[1], r2 = rp[2] - alpha * ap[2];
    rp[0] = r0;
    rp[1] = r1;
    rp[2] = r2;
    double z0, z1, z2;
    minv_apply(V.minv, n, row, r0, r1, r2, z0, z1, z2);
    double* zz = V.z + 3 * (int64_t)row;
    zz[0] = z0;
    zz[1] = z1;
    zz[2] = z2;
    rz += r0 * z0 + r1 * z1 + r2 * z2;
    rr += r0 * r0 + r1 * r1 + r2 * r2;
  }
  rz = block_sum_bcast(rz, red);
  rr = block_sum_bcast(rr, red);
  if (threadIdx.x == 0) {
    part_rz[blockIdx.x] = rz;
    part_rr[blockIdx.x] = rr;
  }
  if (V.fused && blockIdx.x == 0 && threadIdx.x == 0) {
    V.st->pending = 1;
    V.st->started = 1;
  }
}

// End of a slice of fused-update iterations (k_spmv MODE 5): book the last iteration's partials so that the host sees
// its iteration count, residual and convergence flag.  One workgroup.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_book(CgState* st, int parity, const double* __restrict__ part_rz, int n_rz,
                                                const double* __restrict__ part_rr, int n_rr) {
  __shared__ double red[8];
  if (st->done || !st->pending) return;
  const double rz_new = sum_partials_bcast(part_rz, n_rz, red);
  const double rr = sum_partials_bcast(part_rr, n_rr, red);
  if (threadIdx.x == 0) {
    st->rz[parity ^ 1] = rz_new;
    st->rr = rr;
    st->iters += 1;
    st->pending = 0;
    if (rr <= st->tol2) st->done = 1;
  }
}

// beta = rz_new / rz ; p = z + beta p ; workgroup 0 publishes the new scalars
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_update2(CgVec V, int parity, const double* __restrict__ part_rz, int n_rz,
                                                   const double* __restrict__ part_rr, int n_rr) {
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
  const int64_t n3 = 3 * (int64_t)V.n_loc;
  double* p = V.p + PS * (int64_t)V.lo;
  static_assert(PS == 3, "the owned part of the gather vector is contiguous");
  // 16 bytes per lane (the vectors are 8-byte aligned only: p starts at 3 * lo doubles)
  const int64_t n2 = n3 >> 1;
  double2_a8* p2 = reinterpret_cast<double2_a8*>(p);
  const double2_a8* z2 = reinterpret_cast<const double2_a8*>(V.z);
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n2; i += (int64_t)gridDim.x * WG) {
    const double2_a8 zv = __builtin_nontemporal_load(z2 + i), pv = p2[i];
    double2_a8 o;
    o.x = zv.x + beta * pv.x;
    o.y = zv.y + beta * pv.y;
    p2[i] = o;
  }
  if ((n3 & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n3 - 1] = V.z[n3 - 1] + beta * p[n3 - 1];
}

// ------------------------------------------------- block-Jacobi with blocks of B poses
// The preconditioner M = blockdiag(A) over groups of B consecutive poses (nb = 3B unknowns, B <= 32).
// A group's block holds the 3x3 diagonal blocks, the LM diagonal and every off-diagonal block whose two
// poses fall in the group (the odometry chain and short loops).  M is SPD (a principal block-diagonal part
// of an SPD matrix).  Its blocks are inverted EXPLICITLY once per LM iteration (Gauss-Jordan in LDS, no
// pivoting needed for SPD), so applying it inside PCG is a dense, fully parallel mat-vec -- no sequential
// triangular solves in the latency-critical loop.
struct GroupPre {
  const double* ginv;  // [n_groups][nb][nb], symmetric
  int32_t B, nb, nb_pad, n_groups;
};

struct GroupPrepArgs {
  const int32_t* inc_ptr;
  const int32_t* inc_col;
  const double* hoff;
  const double* hd;   // 6 planes
  const double* d2;   // [n_loc x 3]
  double* ginv;
  int32_t n_loc, lo, B, nb, n_groups;
};

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_prepare_groups(GroupPrepArgs A) {
  extern __shared__ double Mall[];  // (WG / L) matrices of nb x (nb + 1)
  // L lanes cooperate on one group: a whole workgroup for big blocks, one wave for small ones (nb <= 24),
  // so that four groups share a workgroup.  Control flow is uniform (same nb everywhere): the barriers
  // below are workgroup-wide even when the waves work on different groups.
  const int nb = A.nb, ld = nb + 1;
  const int L = (nb <= 24) ? 64 : WG;
  const int sub = threadIdx.x / L, lt = threadIdx.x - sub * L, gpw = WG / L;
  double* M = Mall + (size_t)sub * nb * ld;
  const int64_t n = A.n_loc;
  const int n_rounds = (A.n_groups + gpw - 1) / gpw;
  for (int round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    const int g = round * gpw + sub;
    const bool live = g < A.n_groups;
    const int g0 = live ? g * A.B : 0, g1 = live ? min(A.n_loc, g0 + A.B) : 0;  // local rows of the group
    const int valid = 3 * (g1 - g0);
    for (int i = lt; i < nb * ld; i += L) M[i] = 0.0;
    __syncthreads();
    // diagonal 3x3 blocks + LM diagonal; identity on the padding of a short last group
    for (int i = lt; i < nb; i += L) {
      if (i >= valid) {
        M[i * ld + i] = 1.0;
      } else {
        const int row = g0 + i / 3, a = i % 3;
        const int base = i - a;
        for (int b = 0; b < 3; ++b) {
          const int pl = (a == b) ? (a == 0 ? 0 : (a == 1 ? 3 : 5)) : ((a + b == 1) ? 1 : ((a + b == 2) ? 2 : 4));
          M[i * ld + base + b] = A.hd[(int64_t)pl * n + row] + (a == b ? A.d2[3 * (int64_t)row + a] : 0.0);
        }
      }
    }
    __syncthreads();
    // off-diagonal blocks whose column pose lies in the same group
    if (live) {
      const int q0 = A.inc_ptr[g0], q1 = A.inc_ptr[g1];
      for (int q = q0 + lt; q < q1; q += L) {
        const int col = A.inc_col[q] - A.lo;
        if (col >= g0 && col < g1) {
          const int row = upper_row(A.inc_ptr, g0, g1, q);
          // Several edges may join the same two poses: their blocks are adjacent in the row (incidences are sorted by
          // column, then by the caller's edge index).  The FIRST incidence of such a run sums the run in that order and
          // is the only writer of the (row, col) block -- no atomics, so the sum does not depend on the thread schedule.
          if (q > A.inc_ptr[row] && A.inc_col[q - 1] - A.lo == col) continue;
          double acc[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
          const int q_end = A.inc_ptr[row + 1];
          for (int qq = q; qq < q_end && A.inc_col[qq] - A.lo == col; ++qq) {
            double h[9];
            hoff_load(A.hoff, qq, h);
#pragma unroll
            for (int c = 0; c < 9; ++c) acc[c] += h[c];
          }
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) M[(3 * (row - g0) + a) * ld + 3 * (col - g0) + b] += acc[3 * a + b];  // (+= the zero / nothing else writes here)
        }
      }
    }
    __syncthreads();
    // in-place Gauss-Jordan inverse (SPD: no pivoting)
    for (int k = 0; k < nb; ++k) {
      const double pinv = 1.0 / M[k * ld + k];
      __syncthreads();
      for (int j = lt; j < nb; j += L) M[k * ld + j] = (j == k) ? pinv : M[k * ld + j] * pinv;
      __syncthreads();
      for (int e = lt; e < nb * nb; e += L) {
        const int i = e / nb, j = e - i * nb;
        if (i != k && j != k) M[i * ld + j] -= M[i * ld + k] * M[k * ld + j];
      }
      __syncthreads();
      for (int i = lt; i < nb; i += L)
        if (i != k) M[i * ld + k] = -M[i * ld + k] * pinv;
      __syncthreads();
    }
    if (live) {
      double* out = A.ginv + (int64_t)g * nb * nb;
      for (int e = lt; e < nb * nb; e += L) {
        const int i = e / nb, j = e - i * nb;
        out[e] = 0.5 * (M[i * ld + j] + M[j * ld + i]);  // keep it exactly symmetric
      }
    }
    __syncthreads();
  }
}

// z = M^-1 r for the groups of this workgroup: r staged in LDS, one thread per unknown
__device__ __forceinline__ double group_apply(const GroupPre& G, int g, int slot, int k, const double* rb) {
  const double* m = G.ginv + (int64_t)g * G.nb * G.nb + k;
  const double* r = rb + slot * G.nb_pad;
  double z = 0.0;
  for (int j = 0; j < G.nb; ++j) z += m[(int64_t)j * G.nb] * r[j];
  return z;
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_init_g(CgVec V, GroupPre G, const double* __restrict__ b,
                                                  double* __restrict__ part_rz, double* __restrict__ part_bb) {
  __shared__ double rb[WG];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int gpw = WG / G.nb_pad, slot = tid / G.nb_pad, k = tid - slot * G.nb_pad;
  const int64_t n3 = 3 * (int64_t)V.n_loc;
  double rz = 0.0, bb = 0.0;
  for (int gbase = blockIdx.x * gpw; gbase < G.n_groups; gbase += gridDim.x * gpw) {
    const int g = gbase + slot;
    const int64_t idx = (int64_t)g * G.nb + k;
    const bool active = slot < gpw && g < G.n_groups && k < G.nb && idx < n3;
    double r = 0.0;
    if (active) r = b[idx];
    rb[tid] = r;
    __syncthreads();
    if (active) {
      const double z = group_apply(G, g, slot, k, rb);
      V.y[idx] = 0.0;
      V.r[idx] = r;
      V.z[idx] = z;
      V.p[3 * (int64_t)V.lo + idx] = z;
      rz += r * z;
      bb += r * r;
    }
    __syncthreads();
  }
  rz = block_sum_bcast(rz, red);
  bb = block_sum_bcast(bb, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_bb[blockIdx.x] = bb;
  }
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_update1_g(CgVec V, GroupPre G, int parity, const double* __restrict__ part_pap,
                                                     int n_pap, double* __restrict__ part_rz, double* __restrict__ part_rr) {
  __shared__ double rb[WG];
  __shared__ double red[8];
  if (V.st->done) return;
  const int tid = threadIdx.x;
  const double pap = sum_partials_bcast(part_pap, n_pap, red);
  const double alpha = V.st->rz[parity] / pap;
  const int gpw = WG / G.nb_pad, slot = tid / G.nb_pad, k = tid - slot * G.nb_pad;
  const int64_t n3 = 3 * (int64_t)V.n_loc;
  double rz = 0.0, rr = 0.0;
  for (int gbase = blockIdx.x * gpw; gbase < G.n_groups; gbase += gridDim.x * gpw) {
    const int g = gbase + slot;
    const int64_t idx = (int64_t)g * G.nb + k;
    const bool active = slot < gpw && g < G.n_groups && k < G.nb && idx < n3;
    double r = 0.0;
    if (active) {
      V.y[idx] += alpha * V.p[3 * (int64_t)V.lo + idx];
      r = V.r[idx] - alpha * V.ap[idx];
      V.r[idx] = r;
    }
    rb[tid] = r;
    __syncthreads();
    if (active) {
      const double z = group_apply(G, g, slot, k, rb);
      V.z[idx] = z;
      rz += r * z;
      rr += r * r;
    }
    __syncthreads();
  }
  rz = block_sum_bcast(rz, red);
  rr = block_sum_bcast(rr, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_rr[blockIdx.x] = rr;
  }
  if (V.fused && blockIdx.x == 0 && threadIdx.x == 0) {
    V.st->pending = 1;
    V.st->started = 1;
  }
}

// ------------------------------------------------- chain (block-tridiagonal) preconditioner
// Block-Jacobi over segments of L consecutive poses (L = 64 by default; any multiple of 4 that divides 256) whose blocks
// are kept block-TRIDIAGONAL:
//     M_seg = sum of J'J over the edges joining consecutive poses of the segment (the odometry chain)
//             + the 3x3 diagonal blocks of every other edge + D'D          (a sum of PSD terms + D'D: SPD)
// i.e. the block-tridiagonal part of (H + D'D) inside the segment.  Factorised once per LM iteration as block LDL'
//     S_i = M_ii - W_i C_i',   W_i = C_i S_{i-1}^-1,   C_i = H_{i,i-1}      (W_i = 0 at a segment start)
// (k_chain_factor, one thread per segment) and applied as  t_i = r_i - W_i t_{i-1};  z_i = S_i^-1 t_i - W_{i+1}' z_{i+1}.
//
// Apply (chain_apply, fused into the CG update kernels): one wavefront per tile of 256 rows, one lane per CHUNK of 4
// consecutive poses.  Each recurrence is a composition of affine maps x -> a + F x, so a lane (1) sweeps its chunk from
// a zero input, accumulating the chunk's map (a, F = product of its 4 matrices), (2) the 64 chunk maps are combined by
// a Hillis-Steele scan over the wave (6 levels of (3-vector, 3x3) pairs), (3) the lane sweeps its chunk again from its
// true input.  Because W = 0 at every segment start, the chunk maps of different segments do not interact: the scan needs
// no segment masks and the kernel is independent of L.  Work per pose ~150 fp64 FMA and 4.5 64-bit shuffles (the
// lane-per-pose scan this replaces: 380 FMA and 126 shuffles per pose, 76 us per apply at 1M poses, shuffle-bound).
// Factor planes are stored TRANSPOSED inside each 256-row tile (chain_tidx) so that step k of all 64 lanes is one
// coalesced 512-byte access; the vectors go through a wave-private LDS tile (coalesced global access, strided LDS access
// with one pad per chunk: conflict-free).  120 B/pose of factors per PCG iteration (dense 4-pose blocks: 288 B/pose).
constexpr int CHAIN_CHUNK = 4;
constexpr int CHAIN_TILE = 256;                       // rows per wavefront = 64 lanes x CHAIN_CHUNK
constexpr int CHAIN_LDS = CHAIN_TILE * 3 + 64;        // doubles of LDS per wavefront (one pad per chunk)

__host__ __device__ __forceinline__ int64_t chain_tidx(int64_t i) {
  return (i & ~(int64_t)(CHAIN_TILE - 1)) + ((i & (CHAIN_CHUNK - 1)) << 6) + ((i & (CHAIN_TILE - 1)) / CHAIN_CHUNK);
}

// the same for tiles of 64 * ch rows, ch consecutive poses per lane (ch = 4: chain_tidx); used by the lean apply below
__host__ __device__ __forceinline__ int64_t chain_tidx_g(int64_t i, int ch) {
  const int64_t tile = 64 * (int64_t)ch;
  return (i / tile) * tile + ((i % ch) << 6) + ((i % tile) / ch);
}

struct ChainPre {
  const double* cw;   // 9 planes [n_pad]: W_i, row-major 3x3, at chain_tidx(i); 0 at a segment start and in the padding
  const double* cs;   // 6 planes [n_pad]: S_i^-1 (00 01 02 11 12 22), at chain_tidx(i); 0 in the padding
  int32_t n_loc;
  int32_t n_pad;      // n_loc rounded up to whole 256-row tiles
};

// Rows whose block (i, i-1) is the sum of several edges' blocks (listed at create; rare): the C part of their record is
// rewritten as the sum over the run of incidences in incidence order.  One thread per listed row.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_chain_dupfix(const int32_t* __restrict__ rows, int n_rows, const int32_t* __restrict__ inc_ptr,
                               const int32_t* __restrict__ inc_col, const double* __restrict__ hoff, int lo,
                               double* __restrict__ rec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rows) return;
  const int row = rows[k], target = lo + row - 1;
  double acc[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int q = inc_ptr[row]; q < inc_ptr[row + 1]; ++q)
    if (inc_col[q] == target) {
      double h[9];
      hoff_load(hoff, q, h);
#pragma unroll
      for (int c = 0; c < 9; ++c) acc[c] += h[c];
    }
#pragma unroll
  for (int c = 0; c < 9; ++c) rec[(int64_t)row * CHAIN_REC + 6 + c] = acc[c];
}

// One thread per segment, sequential along the chain (seg_len dependent 3x3 steps; once per LM iteration); each step
// reads one 128-byte record.  One wavefront = 64 segments.
//  * The records reach the threads THROUGH LDS: a thread's own records are 128 bytes in a line of their own, so loading
//    them per thread makes every wave instruction touch 64 lines (8 KB of distinct lines per step in flight; the first
//    form of this kernel kept a ring of four such steps per thread and was paced by the L1).  Instead the wavefront
//    loads, for each of its 64 segments, the 1 KB that holds the segment's next eight records with ONE coalesced
//    LDS-DMA instruction (64 lanes x 16 bytes, global_load_lds) into the other half of a double-buffered LDS tile while
//    the current eight steps compute (segment stride 65 x 16 bytes: conflict-free 16-byte reads).
//  * The factor planes are transposed for the APPLY kernels (step k of all 64 lanes = one 512-byte access), which makes
//    this thread's own 15 values per step land 8 bytes apiece in 15 different lines -- 468 MB written for 120 MB of
//    factors at 1M poses.  So the results of GS = 8 consecutive steps are kept in registers and flushed together: in the
//    transposed layout the values of poses j, j + CHUNK, j + 2 CHUNK, ... of one segment are adjacent, i.e. GS / CHUNK
//    consecutive doubles per (plane, k) -- 32-byte (CHUNK = 2) or 16-byte (CHUNK = 4) pieces instead of 8-byte ones.
template <int CHUNK>
__global__ __launch_bounds__(64) void k_chain_factor(const double* __restrict__ rec, int n_loc, int n_pad, int seg_len,
                                                     double* __restrict__ cw, double* __restrict__ cs) {
  constexpr int GS = 8, RUN = GS / CHUNK;   // RUN adjacent doubles per (plane, k) and group
  constexpr int SEGS = 64, STR = GS * 8 + 1;   // double2 per segment and buffer (+ 1: bank spread)
  __shared__ double2 stage[2][SEGS * STR];
  const int lane = threadIdx.x;
  const int64_t n = n_loc, np = n_pad;
  const int64_t seg0 = (int64_t)blockIdx.x * SEGS;
  const int64_t s0 = (seg0 + lane) * seg_len;
  const bool live = s0 < n;
  const int64_t s1 = !live ? s0 : (s0 + seg_len < n ? s0 + seg_len : n);
  const int n_groups = (seg_len + GS - 1) / GS;
  // the records of group g of every segment of this wavefront, straight into LDS (LDS-DMA: no register destination, the
  // wave instruction's 64 x 16 bytes land lane-linear at the segment's 1-KB slot): lane j takes 16 bytes of segment sg's
  // eight records
  auto fetch = [&](int g, int buf) {
#pragma unroll 8
    for (int sg = 0; sg < SEGS; ++sg) {
      int64_t row = (seg0 + sg) * seg_len + (int64_t)g * GS + (lane >> 3);
      row = row < n ? row : n - 1;    // (never used: steps beyond the segment or the shard are skipped below)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rec + row * CHAIN_REC + 2 * (lane & 7)),
                                       (__attribute__((address_space(3))) void*)(&stage[buf][sg * STR]), 16, 0, 0);
    }
  };
  fetch(0, 0);
  __syncthreads();   // (drains the DMA: vmcnt(0) in front of the barrier)
  double p00 = 0.0, p01 = 0.0, p02 = 0.0, p11 = 0.0, p12 = 0.0, p22 = 0.0;  // S_{i-1}^-1
  for (int g = 0; g < n_groups; ++g) {
    const int buf = g & 1;
    if (g + 1 < n_groups) fetch(g + 1, buf ^ 1);   // (buffer buf ^ 1 was last read in group g - 1, a barrier ago)
    const int64_t ib = s0 + (int64_t)g * GS;
    double OUT[GS][15];   // W (9) | S^-1 (6) of the group's steps
#pragma unroll
    for (int k = 0; k < GS; ++k) {
      const int64_t i = ib + k;
      if (live && i < s1) {
        const double2* in = &stage[buf][lane * STR + k * 8];
        const double2 v0 = in[0], v1 = in[1], v2 = in[2], v3 = in[3], v4 = in[4], v5 = in[5], v6 = in[6], v7 = in[7];
        double a00 = v0.x, a01 = v0.y, a02 = v1.x, a11 = v1.y, a12 = v2.x, a22 = v2.y;
        double W[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (i > s0) {
          const double C[9] = {v3.x, v3.y, v4.x, v4.y, v5.x, v5.y, v6.x, v6.y, v7.x};
#pragma unroll
          for (int a = 0; a < 3; ++a) {  // W = C S_{i-1}^-1
            W[3 * a] = C[3 * a] * p00 + C[3 * a + 1] * p01 + C[3 * a + 2] * p02;
            W[3 * a + 1] = C[3 * a] * p01 + C[3 * a + 1] * p11 + C[3 * a + 2] * p12;
            W[3 * a + 2] = C[3 * a] * p02 + C[3 * a + 1] * p12 + C[3 * a + 2] * p22;
          }
          // S = M - W C'  (upper triangle; symmetric in exact arithmetic)
          a00 -= W[0] * C[0] + W[1] * C[1] + W[2] * C[2];
          a01 -= W[0] * C[3] + W[1] * C[4] + W[2] * C[5];
          a02 -= W[0] * C[6] + W[1] * C[7] + W[2] * C[8];
          a11 -= W[3] * C[3] + W[4] * C[4] + W[5] * C[5];
          a12 -= W[3] * C[6] + W[4] * C[7] + W[5] * C[8];
          a22 -= W[6] * C[6] + W[7] * C[7] + W[8] * C[8];
        }
        const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
        const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
        p00 = c00 * id;
        p01 = c01 * id;
        p02 = c02 * id;
        p11 = (a00 * a22 - a02 * a02) * id;
        p12 = (a01 * a02 - a00 * a12) * id;
        p22 = (a00 * a11 - a01 * a01) * id;
#pragma unroll
        for (int c = 0; c < 9; ++c) OUT[k][c] = W[c];
        OUT[k][9] = p00; OUT[k][10] = p01; OUT[k][11] = p02; OUT[k][12] = p11; OUT[k][13] = p12; OUT[k][14] = p22;
      }
    }
    // the next group's records have landed (and every lane is done with this buffer).  BEFORE this group's stores are
    // issued: the barrier's wait covers every outstanding memory operation, and the stores of the previous group have had a
    // whole group's arithmetic to complete, these would not have
    __syncthreads();
    // ---- flush the group
    if (live && ib + GS <= s1 && (ib % GS) == 0) {
      // whole aligned group: pose ib + kk + CHUNK m (m < RUN) sits at chain_tidx_g(ib + kk) + m -- RUN adjacent doubles
#pragma unroll
      for (int kk = 0; kk < CHUNK; ++kk) {
        const int64_t ti = chain_tidx_g(ib + kk, CHUNK);
#pragma unroll
        for (int c = 0; c < 15; ++c) {
          double* dst = (c < 9 ? cw + (int64_t)c * np : cs + (int64_t)(c - 9) * np) + ti;
#pragma unroll
          for (int m = 0; m < RUN; m += 2)
            *reinterpret_cast<double2*>(dst + m) = make_double2(OUT[kk + CHUNK * m][c], OUT[kk + CHUNK * (m + 1)][c]);
        }
      }
    } else if (live) {
#pragma unroll
      for (int k = 0; k < GS; ++k) {
        const int64_t i = ib + k;
        if (i < s1) {
          const int64_t ti = chain_tidx_g(i, CHUNK);
#pragma unroll
          for (int c = 0; c < 9; ++c) cw[(int64_t)c * np + ti] = OUT[k][c];
#pragma unroll
          for (int c = 0; c < 6; ++c) cs[(int64_t)c * np + ti] = OUT[k][9 + c];
        }
      }
    }
  }
}

// a += M b ; then (optionally) M <- M N    (3x3 row-major)
__device__ __forceinline__ void affine_compose(double a[3], double M[9], const double b[3], const double N[9], bool with_m) {
  a[0] += M[0] * b[0] + M[1] * b[1] + M[2] * b[2];
  a[1] += M[3] * b[0] + M[4] * b[1] + M[5] * b[2];
  a[2] += M[6] * b[0] + M[7] * b[1] + M[8] * b[2];
  if (with_m) {
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = M[3 * i] * N[j] + M[3 * i + 1] * N[3 + j] + M[3 * i + 2] * N[6 + j];
#pragma unroll
    for (int c = 0; c < 9; ++c) M[c] = R[c];
  }
}

// In place on the wave-private LDS tile `buf` (CHAIN_LDS doubles): on entry r of the 256 rows starting at `wbase`
// (pose q, component c at q*3 + c + (q >> 2); rows >= n_loc hold 0), on exit z = M^-1 r.  All 64 lanes call it; the
// caller separates it from its own accesses to `buf` with barriers.
__device__ __forceinline__ void chain_apply(const ChainPre& C, int64_t wbase, double* __restrict__ buf) {
  const int lane = threadIdx.x & 63;
  const int64_t np = C.n_pad;
  const double* cwl = C.cw + wbase + lane;  // + c * np + k * 64 : W of this lane's pose k, entry c
  const double* csl = C.cs + wbase + lane;
  double* ch = buf + lane * (3 * CHAIN_CHUNK + 1);  // this lane's chunk: pose k at 3 k
  // The lane's factors are (re)loaded as one batch at the start of each sweep (coalesced; L2 hits after the first):
  // one memory latency per sweep, and only one sweep's factors are live at a time.
  double W[CHAIN_CHUNK][9];
  auto load_w = [&](int shift) {  // shift 0: W_k of the chunk's poses; 1: W_{k+1} (the backward sweeps)
#pragma unroll
    for (int k = 0; k < CHAIN_CHUNK; ++k) {
      if (shift && k == CHAIN_CHUNK - 1) {
        // the pose after the chunk = the next lane's first one (0 for lane 63: the next tile starts a segment)
#pragma unroll
        for (int c = 0; c < 9; ++c) W[k][c] = lane < 63 ? cwl[(int64_t)c * np + 1] : 0.0;
      } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) W[k][c] = cwl[(int64_t)c * np + (k + shift) * 64];
      }
    }
  };
  load_w(0);

  double t[3] = {0.0, 0.0, 0.0}, F[9];
  // ---- forward, pass 1: chunk map from a zero input:  t <- r_k - W_k t ;  F <- (-W_k) F
#pragma unroll
  for (int k = 0; k < CHAIN_CHUNK; ++k) {
    const double* w = W[k];
    const double t0 = ch[3 * k] - (w[0] * t[0] + w[1] * t[1] + w[2] * t[2]);
    const double t1 = ch[3 * k + 1] - (w[3] * t[0] + w[4] * t[1] + w[5] * t[2]);
    const double t2 = ch[3 * k + 2] - (w[6] * t[0] + w[7] * t[1] + w[8] * t[2]);
    t[0] = t0; t[1] = t1; t[2] = t2;
    if (k == 0) {
#pragma unroll
      for (int c = 0; c < 9; ++c) F[c] = -w[c];
    } else {
      double R[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = -(w[3 * i] * F[j] + w[3 * i + 1] * F[3 + j] + w[3 * i + 2] * F[6 + j]);
#pragma unroll
      for (int c = 0; c < 9; ++c) F[c] = R[c];
    }
  }
  // ---- scan of the chunk maps (inclusive): t becomes the true value at the end of this lane's chunk
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double b[3], N[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = __shfl_up(t[k], off, 64);
    if (off < 32) {
#pragma unroll
      for (int c = 0; c < 9; ++c) N[c] = __shfl_up(F[c], off, 64);
    }
    if (lane >= off) affine_compose(t, F, b, N, off < 32);
  }
  // ---- forward, pass 2 from the true input (end of the previous chunk), then u = S^-1 t, stored over r
  {
    double S[CHAIN_CHUNK][6];
#pragma unroll
    for (int k = 0; k < CHAIN_CHUNK; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c) S[k][c] = csl[(int64_t)c * np + k * 64];
    double x0 = __shfl_up(t[0], 1, 64), x1 = __shfl_up(t[1], 1, 64), x2 = __shfl_up(t[2], 1, 64);
    if (lane == 0) { x0 = 0.0; x1 = 0.0; x2 = 0.0; }
#pragma unroll
    for (int k = 0; k < CHAIN_CHUNK; ++k) {
      const double* w = W[k];
      const double* q = S[k];
      const double t0 = ch[3 * k] - (w[0] * x0 + w[1] * x1 + w[2] * x2);
      const double t1 = ch[3 * k + 1] - (w[3] * x0 + w[4] * x1 + w[5] * x2);
      const double t2 = ch[3 * k + 2] - (w[6] * x0 + w[7] * x1 + w[8] * x2);
      x0 = t0; x1 = t1; x2 = t2;
      ch[3 * k] = q[0] * t0 + q[1] * t1 + q[2] * t2;
      ch[3 * k + 1] = q[1] * t0 + q[3] * t1 + q[4] * t2;
      ch[3 * k + 2] = q[2] * t0 + q[4] * t1 + q[5] * t2;
    }
  }
  // ---- backward, pass 1 (k = 3..0) from a zero input:  z <- u_k - W_{k+1}' z ;  F <- (-W_{k+1}') F
  load_w(1);
  t[0] = 0.0; t[1] = 0.0; t[2] = 0.0;
#pragma unroll
  for (int k = CHAIN_CHUNK - 1; k >= 0; --k) {
    const double* w = W[k];
    const double z0 = ch[3 * k] - (w[0] * t[0] + w[3] * t[1] + w[6] * t[2]);
    const double z1 = ch[3 * k + 1] - (w[1] * t[0] + w[4] * t[1] + w[7] * t[2]);
    const double z2 = ch[3 * k + 2] - (w[2] * t[0] + w[5] * t[1] + w[8] * t[2]);
    t[0] = z0; t[1] = z1; t[2] = z2;
    if (k == CHAIN_CHUNK - 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) F[3 * i + j] = -w[3 * j + i];
    } else {
      double R[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[3 * i + j] = -(w[i] * F[j] + w[3 + i] * F[3 + j] + w[6 + i] * F[6 + j]);
#pragma unroll
      for (int c = 0; c < 9; ++c) F[c] = R[c];
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double b[3], N[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) b[k] = __shfl_down(t[k], off, 64);
    if (off < 32) {
#pragma unroll
      for (int c = 0; c < 9; ++c) N[c] = __shfl_down(F[c], off, 64);
    }
    if (lane + off < 64) affine_compose(t, F, b, N, off < 32);
  }
  // ---- backward, pass 2 from the true input (start of the next chunk), z stored over u
  {
    double x0 = __shfl_down(t[0], 1, 64), x1 = __shfl_down(t[1], 1, 64), x2 = __shfl_down(t[2], 1, 64);
    if (lane == 63) { x0 = 0.0; x1 = 0.0; x2 = 0.0; }
#pragma unroll
    for (int k = CHAIN_CHUNK - 1; k >= 0; --k) {
      const double* w = W[k];
      const double z0 = ch[3 * k] - (w[0] * x0 + w[3] * x1 + w[6] * x2);
      const double z1 = ch[3 * k + 1] - (w[1] * x0 + w[4] * x1 + w[7] * x2);
      const double z2 = ch[3 * k + 2] - (w[2] * x0 + w[5] * x1 + w[8] * x2);
      x0 = z0; x1 = z1; x2 = z2;
      ch[3 * k] = z0;
      ch[3 * k + 1] = z1;
      ch[3 * k + 2] = z2;
    }
  }
}

// LDS position of flat element e (= 3 * pose + component) of a wave tile
__device__ __forceinline__ int chain_lds_pos(int e) { return e + e / (3 * CHAIN_CHUNK); }

// PCG start-up with the chain preconditioner: r = b, z = M^-1 r, y = 0, p = z; partials of r.z and b.b.
// One wavefront per 256-row tile; workgroup = 4 waves = 1024 rows.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_init_c(CgVec V, ChainPre C, const double* __restrict__ b,
                                                  double* __restrict__ part_rz, double* __restrict__ part_bb) {
  __shared__ double tile[4][CHAIN_LDS];
  __shared__ double red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* buf = tile[wave];
  const int64_t n3 = 3 * (int64_t)V.n_loc;
  // distinct arrays: let the compiler batch the loads of an unrolled group ahead of its stores
  double* __restrict__ vy = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vz = V.z;
  double* __restrict__ vp = V.p + 3 * (int64_t)V.lo;
  double rz = 0.0, bb = 0.0;
  for (int64_t wg_base = (int64_t)blockIdx.x * (4 * CHAIN_TILE); wg_base < V.n_loc; wg_base += (int64_t)gridDim.x * (4 * CHAIN_TILE)) {
    const int64_t wbase = wg_base + (int64_t)wave * CHAIN_TILE;
    const bool active = wbase < V.n_loc;
    const int64_t f0 = 3 * wbase;
    if (active) {
#pragma unroll 4
      for (int e = lane; e < 3 * CHAIN_TILE; e += 64) {
        const int64_t idx = f0 + e;
        buf[chain_lds_pos(e)] = idx < n3 ? b[idx] : 0.0;
      }
    }
    __syncthreads();
    if (active) chain_apply(C, wbase, buf);
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int e = lane; e < 3 * CHAIN_TILE; e += 64) {
        const int64_t idx = f0 + e;
        if (idx < n3) {
          const double z = buf[chain_lds_pos(e)], r = b[idx];
          vy[idx] = 0.0;
          vr[idx] = r;
          vz[idx] = z;
          vp[idx] = z;
          rz += r * z;
          bb += r * r;
        }
      }
    }
    __syncthreads();
  }
  rz = block_sum_bcast(rz, red);
  bb = block_sum_bcast(bb, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_bb[blockIdx.x] = bb;
  }
}

// x += alpha p ; r -= alpha A p ; z = M^-1 r (chain) ; partials of r.z and r.r
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_cg_update1_c(CgVec V, ChainPre C, int parity, const double* __restrict__ part_pap,
                                                     int n_pap, double* __restrict__ part_rz, double* __restrict__ part_rr) {
  __shared__ double tile[4][CHAIN_LDS];
  __shared__ double red[8];
  if (V.st->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double pap = sum_partials_bcast(part_pap, n_pap, red);
  const double alpha = V.st->rz[parity] / pap;
  double* buf = tile[wave];
  const int64_t n3 = 3 * (int64_t)V.n_loc;
  double* __restrict__ vy = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vz = V.z;
  const double* __restrict__ vap = V.ap;
  const double* __restrict__ pown = V.p + 3 * (int64_t)V.lo;
  double rz = 0.0, rr = 0.0;
  for (int64_t wg_base = (int64_t)blockIdx.x * (4 * CHAIN_TILE); wg_base < V.n_loc; wg_base += (int64_t)gridDim.x * (4 * CHAIN_TILE)) {
    const int64_t wbase = wg_base + (int64_t)wave * CHAIN_TILE;
    const bool active = wbase < V.n_loc;
    const int64_t f0 = 3 * wbase;
    if (active) {
#pragma unroll 4
      for (int e = lane; e < 3 * CHAIN_TILE; e += 64) {
        const int64_t idx = f0 + e;
        double r = 0.0;
        if (idx < n3) {
          vy[idx] += alpha * pown[idx];
          r = vr[idx] - alpha * vap[idx];
          vr[idx] = r;
        }
        buf[chain_lds_pos(e)] = r;
      }
    }
    __syncthreads();
    if (active) chain_apply(C, wbase, buf);
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int e = lane; e < 3 * CHAIN_TILE; e += 64) {
        const int64_t idx = f0 + e;
        if (idx < n3) {
          const double z = buf[chain_lds_pos(e)], r = vr[idx];
          vz[idx] = z;
          rz += r * z;
          rr += r * r;
        }
      }
    }
    __syncthreads();
  }
  rz = block_sum_bcast(rz, red);
  rr = block_sum_bcast(rr, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_rr[blockIdx.x] = rr;
  }
  if (V.fused && blockIdx.x == 0 && threadIdx.x == 0) {
    V.st->pending = 1;
    V.st->started = 1;
  }
}

// ------------------------------------------------- chain preconditioner, lean apply (segments of <= 64 * CH rows)
// Second form of the same apply, built for occupancy: the scan-based kernel above keeps ~244 VGPRs live (2 waves/SIMD)
// and couples its four waves with workgroup barriers, so in the PCG loop -- where its 288 B/row arrive from HBM, the
// SpMV stream having flushed every cache -- it is latency-bound (81 us at 1M poses, 0.44 of the HBM roofline).  Here
//   * a wavefront owns a tile of 64 * CH rows (CH = 2: 128 rows, 30 doubles of factors per lane instead of 60);
//   * the chunk maps are NOT scanned as (vector, matrix) pairs: the chunk's matrix F = (-W_{CH-1}) ... (-W_0) depends on
//     the factors only, so the values at the chunk ends follow from the first-order recurrence x_l = a_l + F_l x_{l-1}
//     over the lanes, run as (lanes per segment - 1) steps of one DPP wave shift (wave_shr:1 / wave_shl:1, bound_ctrl:
//     lane 0 / 63 read 0) + 9 FMA -- no matrix temporaries, no LDS-crossbar shuffles.  F_l = 0 on the first lane of a
//     segment (W = 0 there), so segments do not interact and no masks are needed;
//   * W stays in registers for both sweeps (the backward sweep needs W_{k+1}: the next lane's W_0 arrives by one DPP
//     shift), S^-1 is loaded while the forward recurrence runs: ONE batch of factor loads per tile instead of three;
//   * the four waves of a workgroup never wait for each other inside the tile loop (wave-private LDS tile, wave-scope
//     fences), so their memory phases drift apart and overlap.
// Factor planes: same 9 + 6 planes, transposed inside tiles of 64 * CH rows (chain_tidx_g).

__device__ __forceinline__ void wave_lds_sync() {
  // LDS instructions of one wavefront execute in order; this only stops the compiler from moving them across
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// value of lane - 1 (0 in lane 0) / lane + 1 (0 in lane 63): DPP wavefront shifts, two v_mov_b32_dpp per double
__device__ __forceinline__ double wave_prev(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_next(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// In place on the lane's chunk `ch` of the wave-private LDS tile (pose k, component c at 3 k + c): r on entry, z = M^-1 r
// on exit.  W: this lane's W_k (registers, loaded by the caller).  csl: S^-1 planes at this lane (+ c * np + k * 64).
// n_steps = lanes per segment - 1.
// scan_levels > 0: the recurrence over the lanes runs as that many Hillis-Steele levels of (vector, matrix) pairs instead of
// n_steps serial shifts -- for few, long segments (256 poses = 64 lanes: 6 levels against 63 dependent steps), where
// nothing else is in flight to hide the serial chain (solo.hip.h).
template <int CH, bool NT = true>
__device__ __forceinline__ void chain_apply_lean(const double (&W)[CH][9], const double* __restrict__ cs_tile, int64_t np,
                                                 unsigned lane, double* __restrict__ ch, int n_steps, int scan_levels = 0) {
  // S^-1 (uniform plane base + lane): issued now, consumed after the forward recurrence
  double S[CH][6];
#pragma unroll
  for (int k = 0; k < CH; ++k)
#pragma unroll
    for (int c = 0; c < 6; ++c) S[k][c] = ld_sel<NT>(cs_tile + ((int64_t)c * np + k * 64) + lane);
  // ---- forward, chunk map from a zero input: a = t_{CH-1}, F = (-W_{CH-1}) ... (-W_0)
  double a[3] = {ch[0], ch[1], ch[2]}, F[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) F[c] = -W[0][c];
#pragma unroll
  for (int k = 1; k < CH; ++k) {
    const double* w = W[k];
    const double t0 = ch[3 * k] - (w[0] * a[0] + w[1] * a[1] + w[2] * a[2]);
    const double t1 = ch[3 * k + 1] - (w[3] * a[0] + w[4] * a[1] + w[5] * a[2]);
    const double t2 = ch[3 * k + 2] - (w[6] * a[0] + w[7] * a[1] + w[8] * a[2]);
    a[0] = t0; a[1] = t1; a[2] = t2;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = -(w[3 * i] * F[j] + w[3 * i + 1] * F[3 + j] + w[3 * i + 2] * F[6 + j]);
#pragma unroll
    for (int c = 0; c < 9; ++c) F[c] = R[c];
  }
  // ---- recurrence over the lanes: x_l = a_l + F_l x_{l-1}
  double x0 = a[0], x1 = a[1], x2 = a[2];
  if (scan_levels > 0) {
    double t[3] = {x0, x1, x2};
    for (int lv = 0, off = 1; lv < scan_levels; ++lv, off <<= 1) {
      double bb[3], N[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) bb[k] = __shfl_up(t[k], off, 64);
#pragma unroll
      for (int c = 0; c < 9; ++c) N[c] = __shfl_up(F[c], off, 64);
      if ((int)lane >= off) affine_compose(t, F, bb, N, true);
    }
    x0 = t[0]; x1 = t[1]; x2 = t[2];
  } else {
    for (int s = 0; s < n_steps; ++s) {
      const double p0 = wave_prev(x0), p1 = wave_prev(x1), p2 = wave_prev(x2);
      x0 = a[0] + (F[0] * p0 + F[1] * p1 + F[2] * p2);
      x1 = a[1] + (F[3] * p0 + F[4] * p1 + F[5] * p2);
      x2 = a[2] + (F[6] * p0 + F[7] * p1 + F[8] * p2);
    }
  }
  // ---- forward, true sweep from the end of the previous chunk; u = S^-1 t kept in registers
  double U[CH][3];
  {
    double t0 = wave_prev(x0), t1 = wave_prev(x1), t2 = wave_prev(x2);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const double* w = W[k];
      const double* q = S[k];
      const double n0 = ch[3 * k] - (w[0] * t0 + w[1] * t1 + w[2] * t2);
      const double n1 = ch[3 * k + 1] - (w[3] * t0 + w[4] * t1 + w[5] * t2);
      const double n2 = ch[3 * k + 2] - (w[6] * t0 + w[7] * t1 + w[8] * t2);
      t0 = n0; t1 = n1; t2 = n2;
      U[k][0] = q[0] * t0 + q[1] * t1 + q[2] * t2;
      U[k][1] = q[1] * t0 + q[3] * t1 + q[4] * t2;
      U[k][2] = q[2] * t0 + q[4] * t1 + q[5] * t2;
    }
  }
  // ---- backward: z_k = u_k - V_k' z_{k+1},  V_k = W_{k+1}; V_{CH-1} = the next lane's W_0 (0 after lane 63)
  double VL[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) VL[c] = wave_next(W[0][c]);
  double b[3] = {U[CH - 1][0], U[CH - 1][1], U[CH - 1][2]}, G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[3 * i + j] = -VL[3 * j + i];
#pragma unroll
  for (int k = CH - 2; k >= 0; --k) {
    const double* w = W[k + 1];
    const double z0 = U[k][0] - (w[0] * b[0] + w[3] * b[1] + w[6] * b[2]);
    const double z1 = U[k][1] - (w[1] * b[0] + w[4] * b[1] + w[7] * b[2]);
    const double z2 = U[k][2] - (w[2] * b[0] + w[5] * b[1] + w[8] * b[2]);
    b[0] = z0; b[1] = z1; b[2] = z2;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = -(w[i] * G[j] + w[3 + i] * G[3 + j] + w[6 + i] * G[6 + j]);
#pragma unroll
    for (int c = 0; c < 9; ++c) G[c] = R[c];
  }
  double y0 = b[0], y1 = b[1], y2 = b[2];
  if (scan_levels > 0) {
    double t[3] = {y0, y1, y2};
    for (int lv = 0, off = 1; lv < scan_levels; ++lv, off <<= 1) {
      double bb[3], N[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) bb[k] = __shfl_down(t[k], off, 64);
#pragma unroll
      for (int c = 0; c < 9; ++c) N[c] = __shfl_down(G[c], off, 64);
      if ((int)lane + off < 64) affine_compose(t, G, bb, N, true);
    }
    y0 = t[0]; y1 = t[1]; y2 = t[2];
  } else {
    for (int s = 0; s < n_steps; ++s) {
      const double p0 = wave_next(y0), p1 = wave_next(y1), p2 = wave_next(y2);
      y0 = b[0] + (G[0] * p0 + G[1] * p1 + G[2] * p2);
      y1 = b[1] + (G[3] * p0 + G[4] * p1 + G[5] * p2);
      y2 = b[2] + (G[6] * p0 + G[7] * p1 + G[8] * p2);
    }
  }
  {
    double z0 = wave_next(y0), z1 = wave_next(y1), z2 = wave_next(y2);
#pragma unroll
    for (int k = CH - 1; k >= 0; --k) {
      double VN[9];  // shifted again instead of kept live through the recurrence
      if (k == CH - 1) {
#pragma unroll
        for (int c = 0; c < 9; ++c) VN[c] = wave_next(W[0][c]);
      }
      const double* w = (k == CH - 1) ? VN : W[(k + 1 < CH) ? k + 1 : 0];
      const double n0 = U[k][0] - (w[0] * z0 + w[3] * z1 + w[6] * z2);
      const double n1 = U[k][1] - (w[1] * z0 + w[4] * z1 + w[7] * z2);
      const double n2 = U[k][2] - (w[2] * z0 + w[5] * z1 + w[8] * z2);
      z0 = n0; z1 = n1; z2 = n2;
      ch[3 * k] = z0;
      ch[3 * k + 1] = z1;
      ch[3 * k + 2] = z2;
    }
  }
}

// PCG start-up, lean chain apply: r = b, z = M^-1 r, y = 0, p = z; partials of r.z and b.b.  One wavefront per tile of
// 64 * CH rows, tiles dealt round-robin over all waves of the grid.  FULL tiles (all but possibly the last) run without
// per-element predicates: every load of the tile is issued before the first use.
// NW = wavefronts per workgroup: 4 on large graphs; 1 on small ones, where four 74-KB tiles behind ONE compute unit's L1
// took 3.9 us to load (INTEL, 5 tiles) and a workgroup per tile spreads them over the chip.
template <int NW>
__device__ __forceinline__ double block_sum_nw(double v, double* sh) {
  if constexpr (NW == 4) {
    return block_sum_bcast(v, sh);
  } else {
    return __shfl(wave_sum(v), 0, 64);
  }
}
template <int CH, int NW>
__global__ __launch_bounds__(64 * NW) void k_cg_init_cl(CgVec V, ChainPre C, int n_steps, int scan_levels, const double* __restrict__ b,
                                                   double* __restrict__ part_rz, double* __restrict__ part_bb) {
  constexpr int TILE = 64 * CH, STRIDE = 3 * CH + 1, NV = 3 * CH;
  __shared__ double tile[NW][64 * STRIDE];
  __shared__ double red[8];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  double* buf = tile[wave];
  double* ch = buf + lane * STRIDE;
  const int64_t n3 = 3 * (int64_t)V.n_loc, np = C.n_pad;
  double* __restrict__ vy = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vz = V.z;
  double* __restrict__ vp = V.p + 3 * (int64_t)V.lo;
  double rz = 0.0, bb = 0.0;
  const int64_t n_tiles = ((int64_t)V.n_loc + TILE - 1) / TILE;
  for (int64_t t = (int64_t)blockIdx.x * NW + wave; t < n_tiles; t += (int64_t)gridDim.x * NW) {
    const int64_t wbase = t * TILE, f0 = 3 * wbase;   // uniform: plane / vector bases stay scalar, lane offsets 32-bit
    const unsigned lim = (unsigned)(n3 - f0 < 3 * TILE ? n3 - f0 : 3 * TILE);
    const double* cw_tile = C.cw + wbase;
    double W[CH][9];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int c = 0; c < 9; ++c) W[k][c] = ld_stream(cw_tile + ((int64_t)c * np + k * 64) + lane);
    const double* bt = b + f0;
    double rv[NV];
    if (lim == 3u * TILE) {
#pragma unroll
      for (int j = 0; j < NV; ++j) rv[j] = bt[lane + 64u * j];
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        const double v = bt[e < lim ? e : 0u];   // clamped address: no load behind a branch
        rv[j] = e < lim ? v : 0.0;
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const unsigned e = lane + 64u * j;
      buf[e + e / (3 * CH)] = rv[j];
    }
    wave_lds_sync();
    chain_apply_lean<CH>(W, C.cs + wbase, np, lane, ch, n_steps, scan_levels);
    wave_lds_sync();
    double* yt = vy + f0;
    double* rt = vr + f0;
    double* zt = vz + f0;
    double* pt = vp + f0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const unsigned e = lane + 64u * j;
      const double z = buf[e + e / (3 * CH)], r = rv[j];
      if (lim == 3u * TILE || e < lim) {
        yt[e] = 0.0;
        rt[e] = r;
        zt[e] = z;
        pt[e] = z;
      }
      rz += r * z;   // rows past the end hold r = z = 0
      bb += r * r;
    }
    wave_lds_sync();
  }
  rz = block_sum_nw<NW>(rz, red);
  bb = block_sum_nw<NW>(bb, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_bb[blockIdx.x] = bb;
  }
}

// x += alpha p ; r -= alpha A p ; z = M^-1 r (lean chain apply) ; partials of r.z and r.r
#ifdef PGO_PHASE_TIMING
static __device__ unsigned long long g_phase_t[16];
#define PGO_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_t[k] = wall_clock64(); } while (0)
#else
#define PGO_T(k) do { } while (0)
#endif
template <int CH, int NW>
__global__ __launch_bounds__(64 * NW) void k_cg_update1_cl(CgVec V, ChainPre C, int n_steps, int scan_levels, int parity,
                                                      const double* __restrict__ part_pap, int n_pap,
                                                      double* __restrict__ part_rz, double* __restrict__ part_rr) {
  constexpr int TILE = 64 * CH, STRIDE = 3 * CH + 1, NV = 3 * CH;
  __shared__ double tile[NW][64 * STRIDE];
  __shared__ double red[8];
  PGO_T(0);
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  double alpha = 0.0, pap_part = 0.0;
  if constexpr (NW == 4) {
    if (V.st->done) return;
    const double pap = sum_partials_bcast(part_pap, n_pap, red);
    alpha = V.st->rz[parity] / pap;
  } else {
    // small graphs: the flag, r.z and the partials leave together (one round trip instead of three); the sum itself
    // waits until the tile's own loads have been issued
    const int dn = V.st->done;
    alpha = V.st->rz[parity];
    for (int i = threadIdx.x; i < n_pap; i += 64) pap_part += part_pap[i];
    if (dn) return;
  }
  PGO_T(1);
  PGO_T(2);
  double* buf = tile[wave];
  double* ch = buf + lane * STRIDE;
  const int64_t n3 = 3 * (int64_t)V.n_loc, np = C.n_pad;
  double* __restrict__ vy = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vz = V.z;
  const double* __restrict__ vap = V.ap;
  const double* __restrict__ pown = V.p + 3 * (int64_t)V.lo;
  double rz = 0.0, rr = 0.0;
  const int64_t n_tiles = ((int64_t)V.n_loc + TILE - 1) / TILE;
  for (int64_t t = (int64_t)blockIdx.x * NW + wave; t < n_tiles; t += (int64_t)gridDim.x * NW) {
    const int64_t wbase = t * TILE, f0 = 3 * wbase;   // uniform: plane / vector bases stay scalar, lane offsets 32-bit
    const unsigned lim = (unsigned)(n3 - f0 < 3 * TILE ? n3 - f0 : 3 * TILE);
    const bool full = lim == 3u * TILE;
    const double* cw_tile = C.cw + wbase;
    double W[CH][9];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int c = 0; c < 9; ++c) W[k][c] = ld_stream(cw_tile + ((int64_t)c * np + k * 64) + lane);
    double* yt = vy + f0;
    double* rt = vr + f0;
    const double* at = vap + f0;
    const double* pt = pown + f0;
    double rv[NV], yv[NV];
    if (full) {
      double av[NV], pv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        rv[j] = ld_stream(rt + e);
        av[j] = ld_stream(at + e);
        yv[j] = ld_stream(yt + e);
        pv[j] = pt[e];
      }
      if constexpr (NW == 1) {
        if (t == (int64_t)blockIdx.x) alpha = alpha / block_sum_nw<NW>(pap_part, red);  // first tile of this wave
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        rv[j] -= alpha * av[j];
        yv[j] += alpha * pv[j];
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        st_stream(rt + e, rv[j]);
        st_stream(yt + e, yv[j]);
      }
    } else {
      // the last, partial tile: the same batched loads from clamped addresses (element 0 of the tile stands in for the
      // elements past the end), values masked, stores predicated -- no load sits behind a branch (on the small graphs this
      // tile is a fifth of the kernel: INTEL 13.9 us per launch with per-element branches and their waits)
      double av[NV], pv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j, ec = e < lim ? e : 0u;
        rv[j] = rt[ec];
        av[j] = at[ec];
        yv[j] = yt[ec];
        pv[j] = pt[ec];
      }
      if constexpr (NW == 1) {
        if (t == (int64_t)blockIdx.x) alpha = alpha / block_sum_nw<NW>(pap_part, red);  // first tile of this wave
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        const bool ok = e < lim;
        rv[j] = ok ? rv[j] - alpha * av[j] : 0.0;
        yv[j] += alpha * pv[j];
        if (ok) {
          rt[e] = rv[j];
          yt[e] = yv[j];
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
    PGO_T(3);
    chain_apply_lean<CH>(W, C.cs + wbase, np, lane, ch, n_steps, scan_levels);
    wave_lds_sync();
    PGO_T(4);
    double* zt = vz + f0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const unsigned e = lane + 64u * j;
      const double z = buf[e + e / (3 * CH)];
      if (full || e < lim) st_stream(zt + e, z);
      rz += rv[j] * z;
    }
    wave_lds_sync();
  }
  PGO_T(5);
  rz = block_sum_nw<NW>(rz, red);
  rr = block_sum_nw<NW>(rr, red);
  if (tid == 0) {
    part_rz[blockIdx.x] = rz;
    part_rr[blockIdx.x] = rr;
  }
  if (V.fused && blockIdx.x == 0 && threadIdx.x == 0) {
    V.st->pending = 1;
    V.st->started = 1;
  }
  PGO_T(6);
}

// ------------------------------------------------- single-reduction PCG (Chronopoulos & Gear), chain preconditioner
// Several ranks: the textbook loop has two dependent reduction points per iteration (p.Ap, then r.z / r.r), i.e. two
// latency-bound all-reduces.  This form carries s = A p by recurrence, so that ONE reduction per iteration suffices:
//     u = M^-1 r,  w = A u,  gamma = r.u,  delta = w.u,  rr = r.r            (all three reduced together)
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)
//     p = u + beta p,  s = w + beta s,  x += alpha p,  r -= alpha s
// The iterates equal the textbook ones in exact arithmetic; the recurrence for s drifts from A p over thousands of
// iterations on the ill-conditioned systems of the exact mode (measured in round 2: MIT at radius 3e11 stagnates above
// 1e-10), so the host takes this loop only for pcg_rtol >= 1e-6.
// Kernels per iteration: k_cg_sr_cl (everything above on the vectors + the chain apply + partials of gamma, rr), the
// exchange of u, the product w = A u with its delta partial (K3), k_finalize + ONE all-reduce of (gamma, rr, delta),
// k_cg_sr_scal.  u lives in the gather vector (V.p, global indexing), p in V.z, s in `sv`, w in V.ap.

// scal = (gamma, rr, delta) reduced over workgroups and ranks; first != 0: start of a solve (scal[1] = b.b)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_cg_sr_scal(CgState* st, const double* __restrict__ scal, double rtol, int first) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double gamma = scal[0], rr = scal[1], delta = scal[2];
  if (first) {
    st->bb = rr;
    st->tol2 = rtol * rtol * rr;
    st->iters = 0;
    st->pending = 0;
    st->started = 0;
    st->done = (rr == 0.0) ? 1 : 0;
    st->rr = rr;
    st->sr_beta = 0.0;
    st->sr_alpha = gamma / delta;
    st->sr_gamma = gamma;
    st->rz[0] = st->rz[1] = gamma;
    return;
  }
  if (st->done) return;
  st->iters += 1;          // the vector update that produced this residual
  st->rr = rr;
  if (rr <= st->tol2) {
    st->done = 1;
    return;
  }
  const double beta = gamma / st->sr_gamma;
  st->sr_alpha = gamma / (delta - beta * gamma / st->sr_alpha);
  st->sr_beta = beta;
  st->sr_gamma = gamma;
}

template <int CH, int NW>
__global__ __launch_bounds__(64 * NW) void k_cg_sr_cl(CgVec V, ChainPre C, double* __restrict__ sv, int n_steps, int scan_levels,
                                                 double* __restrict__ part_gamma, double* __restrict__ part_rr) {
  constexpr int TILE = 64 * CH, STRIDE = 3 * CH + 1, NV = 3 * CH;
  __shared__ double tile[NW][64 * STRIDE];
  __shared__ double red[8];
  const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (V.st->done) return;       // (only k_cg_sr_scal writes the state, never during this launch)
  const double alpha = V.st->sr_alpha, beta = V.st->sr_beta;
  double* buf = tile[wave];
  double* ch = buf + lane * STRIDE;
  const int64_t n3 = 3 * (int64_t)V.n_loc, np = C.n_pad;
  double* __restrict__ vx = V.y;
  double* __restrict__ vr = V.r;
  double* __restrict__ vp = V.z;
  const double* __restrict__ vw = V.ap;
  double* __restrict__ vu = V.p + 3 * (int64_t)V.lo;
  double gam = 0.0, rr = 0.0;
  const int64_t n_tiles = ((int64_t)V.n_loc + TILE - 1) / TILE;
  for (int64_t t = (int64_t)blockIdx.x * NW + wave; t < n_tiles; t += (int64_t)gridDim.x * NW) {
    const int64_t wbase = t * TILE, f0 = 3 * wbase;
    const unsigned lim = (unsigned)(n3 - f0 < 3 * TILE ? n3 - f0 : 3 * TILE);
    const double* cw_tile = C.cw + wbase;
    double W[CH][9];
#pragma unroll
    for (int k = 0; k < CH; ++k)
#pragma unroll
      for (int c = 0; c < 9; ++c) W[k][c] = ld_stream(cw_tile + ((int64_t)c * np + k * 64) + lane);
    double rv[NV];
    {
      double uv[NV], wv[NV], pv[NV], sq[NV], xv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {   // clamped addresses: no load behind a branch (the partial last tile)
        const unsigned e = lane + 64u * j, ec = e < lim ? e : 0u;
        uv[j] = vu[f0 + ec];
        wv[j] = ld_stream(vw + f0 + ec);
        pv[j] = ld_stream(vp + f0 + ec);
        sq[j] = ld_stream(sv + f0 + ec);
        xv[j] = ld_stream(vx + f0 + ec);
        rv[j] = ld_stream(vr + f0 + ec);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const unsigned e = lane + 64u * j;
        const bool ok = e < lim;
        const double pn = uv[j] + beta * pv[j], sn = wv[j] + beta * sq[j];
        const double xn = xv[j] + alpha * pn;
        rv[j] = ok ? rv[j] - alpha * sn : 0.0;
        if (ok) {
          st_stream(vp + f0 + e, pn);
          st_stream(sv + f0 + e, sn);
          st_stream(vx + f0 + e, xn);
          st_stream(vr + f0 + e, rv[j]);
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
    chain_apply_lean<CH>(W, C.cs + wbase, np, lane, ch, n_steps, scan_levels);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const unsigned e = lane + 64u * j;
      const double u = buf[e + e / (3 * CH)];
      if (e < lim) vu[f0 + e] = u;      // plain store: the next kernel gathers it
      gam += rv[j] * u;
    }
    wave_lds_sync();
  }
  gam = block_sum_nw<NW>(gam, red);
  rr = block_sum_nw<NW>(rr, red);
  if (tid == 0) {
    part_gamma[blockIdx.x] = gam;
    part_rr[blockIdx.x] = rr;
  }
}

// ------------------------------------------------- METHOD 2: switch variables, eliminated edge by edge
// A switch s_e appears in exactly two residual blocks (its edge and its prior), so it is eliminated from the LM
// system exactly (Schur complement per edge).  With sigma = Jacobi scale of the switch column, j = d e / d s,
//     h_ss = sigma^2 (|j|^2 + lambda),  D_s^2 = clamp(h_ss) / radius,  den = h_ss + D_s^2,
//     g_s  = sigma (j.r - sqrt(lambda) q)            (scaled gradient w.r.t. the switch; q = sqrt(lambda)(1 - s))
//     c = sigma^2 / den,  gamma = sigma g_s / den
// the pose system sees the edge as  J'(I - c j j')J  and  J'(r - gamma j)  (applied in k_assemble), and after the solve
//     t = j'(J S y),  y_s = (g_s - sigma t) / den,  s_cand = s - sigma y_s.
struct SwitchArrays {
  const uint8_t* flags;
  int32_t n_edges;
  double lambda;
  double* sw;      // current switches
  double* cand;    // candidate switches
  double* js;      // [n x 3]
  double* sigma;   // Jacobi scale of the switch column
  double* c;
  double* gamma;
  double* gs;      // scaled gradient g_s
  double* den;
  double* hss;
};

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_switch_scale(SwitchArrays W, int enabled) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= W.n_edges) return;
  double sg = 1.0;
  if ((W.flags[e] & 1u) && enabled) {
    const double* j = W.js + 3 * (int64_t)e;
    sg = 1.0 / (1.0 + sqrt(j[0] * j[0] + j[1] * j[1] + j[2] * j[2] + W.lambda));
  }
  W.sigma[e] = sg;
}

// per LM iteration: elimination coefficients; partial max of the unscaled switch gradient and of sum s^2
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_switch_prepare(SwitchArrays W, const double* __restrict__ jr, double radius, double dmin,
                                                       double dmax, double* __restrict__ part_gmax,
                                                       double* __restrict__ part_s2) {
  __shared__ double red[8];
  double gm = 0.0, s2 = 0.0;
  for (int e = blockIdx.x * WG + threadIdx.x; e < W.n_edges; e += gridDim.x * WG) {
    const unsigned fl = W.flags[e];
    if (!(fl & 1u)) {
      W.c[e] = 0.0;
      W.gamma[e] = 0.0;
      continue;
    }
    const double* j = W.js + 3 * (int64_t)e;
    const double* r = jr + (int64_t)e * REC + 10;
    const double sv = W.sw[e], sg = W.sigma[e];
    const double q = sqrt(W.lambda) * (1.0 - sv);
    const double gun = j[0] * r[0] + j[1] * r[1] + j[2] * r[2] - sqrt(W.lambda) * q;  // unscaled gradient w.r.t. s
    const double hss = sg * sg * (j[0] * j[0] + j[1] * j[1] + j[2] * j[2] + W.lambda);
    const double den = hss + fmin(fmax(hss, dmin), dmax) / radius;
    W.hss[e] = hss;
    W.den[e] = den;
    W.gs[e] = sg * gun;
    W.c[e] = sg * sg / den;
    W.gamma[e] = sg * sg * gun / den;
    if (fl & 2u) {
      gm = fmax(gm, fabs(gun));
      s2 += sv * sv;
    }
  }
  gm = block_max_bcast(gm, red);
  s2 = block_sum_bcast(s2, red);
  if (threadIdx.x == 0) {
    part_gmax[blockIdx.x] = gm;
    part_s2[blockIdx.x] = s2;
  }
}

// after the pose solve: back-substitute the switches; partials of the model-decrease terms and of the step norm
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_switch_backsub(SwitchArrays W, const int32_t* __restrict__ ia, const int32_t* __restrict__ ib,
                                                       const double* __restrict__ jr, const double* __restrict__ scale,
                                                       const double* __restrict__ yfull, double* __restrict__ part_model,
                                                       double* __restrict__ part_step2) {
  __shared__ double red[8];
  double pm = 0.0, ps = 0.0;
  for (int e = blockIdx.x * WG + threadIdx.x; e < W.n_edges; e += gridDim.x * WG) {
    const unsigned fl = W.flags[e];
    if (!(fl & 1u)) {
      W.cand[e] = W.sw[e];
      continue;
    }
    const double* R = jr + (int64_t)e * REC;
    const double* j = W.js + 3 * (int64_t)e;
    const int64_t a = ia[e], b = ib[e];
    // (J S y): row k = sum_c A[k][c] sa[c] ya[c] + B[k][c] sb[c] yb[c],  B = [-A[:,0] | -A[:,1] | (0,0,g2)']
    const double ua0 = scale[3 * a] * yfull[PS * a], ua1 = scale[3 * a + 1] * yfull[PS * a + 1], ua2 = scale[3 * a + 2] * yfull[PS * a + 2];
    const double ub0 = scale[3 * b] * yfull[PS * b], ub1 = scale[3 * b + 1] * yfull[PS * b + 1], ub2 = scale[3 * b + 2] * yfull[PS * b + 2];
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double row = R[3 * k] * (ua0 - ub0) + R[3 * k + 1] * (ua1 - ub1) + R[3 * k + 2] * ua2 + ((k == 2) ? R[9] * ub2 : 0.0);
      t += j[k] * row;
    }
    const double sg = W.sigma[e], gs = W.gs[e], den = W.den[e];
    const double ys = (gs - sg * t) / den;
    W.cand[e] = W.sw[e] - sg * ys;
    if (fl & 2u) {
      // joint model decrease = (reduced pose part) + gamma t + y_s g_s - (c t^2 + 2 y_s sigma t + h_ss y_s^2) / 2
      pm += W.gamma[e] * t + ys * gs - 0.5 * (W.c[e] * t * t + 2.0 * ys * sg * t + W.hss[e] * ys * ys);
      ps += sg * ys * sg * ys;
    }
  }
  pm = block_sum_bcast(pm, red);
  ps = block_sum_bcast(ps, red);
  if (threadIdx.x == 0) {
    part_model[blockIdx.x] = pm;
    part_step2[blockIdx.x] = ps;
  }
}

// ------------------------------------------------- batched handles: per-problem reductions and acceptance
struct ProbRange {
  int32_t row0, nrows;   // rows [row0, row0 + nrows)
  int32_t e0, e1;        // local edges [e0, e1)
};
struct ProbSums {
  double cost, gmax, xnorm2;
};
// one workgroup per problem, fixed summation order (thread-strided partials, then the workgroup tree)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_prob_reduce(const ProbRange* __restrict__ pr, const double* __restrict__ edge_cost,
                                                    const double* __restrict__ gs, const double* __restrict__ scale,
                                                    const double* __restrict__ x, int lo, ProbSums* __restrict__ out) {
  __shared__ double red[8];
  const ProbRange P = pr[blockIdx.x];
  double c = 0.0, gm = 0.0, xn = 0.0;
  if (edge_cost)
    for (int e = P.e0 + threadIdx.x; e < P.e1; e += WG) c += edge_cost[e];
  const int64_t f0 = 3 * (int64_t)P.row0, fn = 3 * (int64_t)P.nrows;
  for (int64_t i = threadIdx.x; i < fn; i += WG) {
    const double sc = scale[3 * (int64_t)lo + f0 + i];
    if (sc > 0.0) {
      if (gs) gm = fmax(gm, fabs(gs[f0 + i] / sc));
      const double xv = x[3 * (int64_t)lo + f0 + i];
      xn += xv * xv;
    }
  }
  c = block_sum_bcast(c, red);
  gm = block_max_bcast(gm, red);
  xn = block_sum_bcast(xn, red);
  if (threadIdx.x == 0) {
    out[blockIdx.x].cost = c;
    out[blockIdx.x].gmax = gm;
    out[blockIdx.x].xnorm2 = xn;
  }
}
// x <- cand on the rows of the accepted problems
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_accept_rows(int n_loc, int lo, const int32_t* __restrict__ prob_of_256, const int32_t* __restrict__ accept,
                              const double* __restrict__ cand, double* __restrict__ x) {
  const int64_t n3 = 3 * (int64_t)n_loc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / 3);
    if (accept[prob_of_256[row >> 8]]) x[3 * (int64_t)lo + i] = cand[3 * (int64_t)lo + i];
  }
}

// Model decrease without a product by H: PCG leaves r = b - (H + D) y (its recurrence residual), so
//     y.(H y) = y.b - y.r - y.(D y)
// -- three dot products over vectors that are there anyway instead of one more SpMV per LM iteration.  Partials of
// y.b, y.r and y.(D y) per workgroup.
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_model_terms(int64_t n, const double* __restrict__ y, const double* __restrict__ b,
                                                    const double* __restrict__ r, const double* __restrict__ d2,
                                                    double* __restrict__ part_yb, double* __restrict__ part_yr,
                                                    double* __restrict__ part_ydy) {
  __shared__ double red[8];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) {
    const double yi = y[i];
    s0 += yi * b[i];
    s1 += yi * r[i];
    s2 += yi * yi * d2[i];
  }
  s0 = block_sum_bcast(s0, red);
  s1 = block_sum_bcast(s1, red);
  s2 = block_sum_bcast(s2, red);
  if (threadIdx.x == 0) {
    part_yb[blockIdx.x] = s0;
    part_yr[blockIdx.x] = s1;
    part_ydy[blockIdx.x] = s2;
  }
}

// partials of a.b over n doubles
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_dot(int64_t n, const double* __restrict__ a, const double* __restrict__ b,
                                            double* __restrict__ part) {
  __shared__ double red[8];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * WG) s += a[i] * b[i];
  s = block_sum_bcast(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_flag_to_double(const int* __restrict__ flag, double* __restrict__ out) { out[0] = (double)(*flag); }

// copy the owned part of a local [n x 3] vector into the padded, globally indexed gather vector
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_scatter_owned(int n_loc, int lo, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t n3 = 3 * (int64_t)n_loc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / 3;
    dst[PS * (lo + row) + (i - 3 * row)] = src[i];
  }
}

// |b - A y|^2 and |b|^2 as partials (test hook "verify_residual": the TRUE residual of a finished PCG solve next to the
// recurrence residual the loop stopped on)
template <int PGO_UNIT_ = 0>
__global__ __launch_bounds__(WG) void k_residual_norm(int64_t n3, const double* __restrict__ b, const double* __restrict__ ay,
                                                      double* __restrict__ part_rr, double* __restrict__ part_bb) {
  __shared__ double red[8];
  double rr = 0.0, bb = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n3; i += (int64_t)gridDim.x * WG) {
    const double d = b[i] - ay[i];
    rr += d * d;
    bb += b[i] * b[i];
  }
  rr = block_sum_bcast(rr, red);
  bb = block_sum_bcast(bb, red);
  if (threadIdx.x == 0) {
    part_rr[blockIdx.x] = rr;
    part_bb[blockIdx.x] = bb;
  }
}

// halo exchange helpers: pack rows of the gather vector into a contiguous buffer / scatter them back
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_pack_rows(int64_t n_rows, const int32_t* __restrict__ rows, const double* __restrict__ src,
                            double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / 3;
    dst[i] = src[PS * (int64_t)rows[k] + (i - 3 * k)];
  }
}
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ void k_unpack_rows(int64_t n_rows, const int32_t* __restrict__ rows, const double* __restrict__ src,
                              double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / 3;
    dst[PS * (int64_t)rows[k] + (i - 3 * k)] = src[i];
  }
}

// candidate = x - S y on the owned rows; partials of |step|^2
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_candidate(int n_loc, int lo, const double* __restrict__ x,
                                                  const double* __restrict__ scale, const double* __restrict__ y,
                                                  double* __restrict__ cand, double* __restrict__ part_step2) {
  __shared__ double red[8];
  double s2 = 0.0;
  const int64_t n3 = 3 * (int64_t)n_loc, off = 3 * (int64_t)lo;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n3; i += (int64_t)gridDim.x * WG) {
    const double d = -scale[off + i] * y[i];
    cand[off + i] = x[off + i] + d;
    s2 += d * d;
  }
  s2 = block_sum_bcast(s2, red);
  if (threadIdx.x == 0) part_step2[blockIdx.x] = s2;
}

// partials of |x|^2 over the owned free parameters (scale == 0 marks the constant pose)
template <int PGO_UNIT_ = 0>   // (a template so that only the translation units that launch it carry it)
__global__ __launch_bounds__(WG) void k_xnorm(int n_loc, int lo, const double* __restrict__ x,
                                              const double* __restrict__ scale, double* __restrict__ part) {
  __shared__ double red[8];
  double s2 = 0.0;
  const int64_t n3 = 3 * (int64_t)n_loc, off = 3 * (int64_t)lo;
  for (int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x; i < n3; i += (int64_t)gridDim.x * WG)
    if (scale[off + i] > 0.0) s2 += x[off + i] * x[off + i];
  s2 = block_sum_bcast(s2, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s2;
}


}  // namespace dev
}  // namespace pgo
