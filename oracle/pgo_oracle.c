/*
 * pgo_oracle.c -- CPU restatement of the reference's DCS pose-graph path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libpgo.so) never does.
 *
 * PARITY UNPINNED: the reference (wei-ght/toy-robust-backend-slam, DCS-ceres/)
 * ships no tests, golden vectors or recorded outputs for this path, and it cannot
 * be built here (needs Ceres, Eigen3, Boost -- none installed, no network).  This
 * file restates the reference's arithmetic from its sources; the Ceres-side policy
 * (loss corrector, LM trust-region rules) is restated from the published Ceres
 * Solver 2.x algorithm (version unpinned by the reference: CMakeLists.txt:9).
 *
 * What follows what (paths relative to /root/reference/DCS-ceres):
 *   jet arithmetic                  ceres::Jet<double,6> as instantiated by
 *                                   AutoDiffCostFunction<...,3,3,3> (src/ceres_error.cpp:34,127)
 *   edge_functor()                  OdometryResidue::operator()  src/ceres_error.cpp:42-94
 *                                   DCSClosureResidue::operator() src/ceres_error.cpp:135-196
 *   mat3_inverse()                  Eigen's fixed 3x3 inverse (cofactors / determinant), used at :87,:180
 *   huber()                         ceres::HuberLoss(0.01)       main.cpp:66-68
 *   corrector                       Ceres Corrector, rho'' <= 0 branch
 *   pgo_oracle_eval()               ResidualBlock::Evaluate over main.cpp:95-150's blocks
 *   pgo_oracle_lm_pcg()             ceres::Solve (main.cpp:154-163) with the linear solve done by
 *                                   block-Jacobi PCG -- the same algorithm the HIP backend runs
 *                                   ("port" CPU baseline); the direct-solve variant that stands in
 *                                   for SPARSE_NORMAL_CHOLESKY lives in oracle/oracle.py.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ jets */
#define NJ 6
typedef struct {
  double a;
  double v[NJ];
} jet;

static jet jc(double c) {
  jet r;
  r.a = c;
  for (int i = 0; i < NJ; ++i) r.v[i] = 0.0;
  return r;
}
static jet jvar(double c, int k) {
  jet r = jc(c);
  r.v[k] = 1.0;
  return r;
}
static jet jadd(jet x, jet y) {
  jet r;
  r.a = x.a + y.a;
  for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] + y.v[i];
  return r;
}
static jet jsub(jet x, jet y) {
  jet r;
  r.a = x.a - y.a;
  for (int i = 0; i < NJ; ++i) r.v[i] = x.v[i] - y.v[i];
  return r;
}
static jet jneg(jet x) {
  jet r;
  r.a = -x.a;
  for (int i = 0; i < NJ; ++i) r.v[i] = -x.v[i];
  return r;
}
static jet jmul(jet x, jet y) {
  jet r;
  r.a = x.a * y.a;
  for (int i = 0; i < NJ; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a;
  return r;
}
static jet jdiv(jet x, jet y) { /* Ceres: g.a_inverse, f.a * g.a_inverse, (f.v - f_by_g * g.v) * g.a_inverse */
  jet r;
  double inv = 1.0 / y.a;
  double q = x.a * inv;
  r.a = q;
  for (int i = 0; i < NJ; ++i) r.v[i] = (x.v[i] - q * y.v[i]) * inv;
  return r;
}
static jet jsin(jet x) {
  jet r;
  double c = cos(x.a);
  r.a = sin(x.a);
  for (int i = 0; i < NJ; ++i) r.v[i] = c * x.v[i];
  return r;
}
static jet jcos(jet x) {
  jet r;
  double s = -sin(x.a);
  r.a = cos(x.a);
  for (int i = 0; i < NJ; ++i) r.v[i] = s * x.v[i];
  return r;
}
static jet jasin(jet x) { /* d asin = 1/sqrt(1 - a^2) */
  jet r;
  double t = 1.0 / sqrt(1.0 - x.a * x.a);
  r.a = asin(x.a);
  for (int i = 0; i < NJ; ++i) r.v[i] = t * x.v[i];
  return r;
}
static jet jsqrt(jet x) {
  jet r;
  double t = sqrt(x.a);
  double h = 1.0 / (2.0 * t);
  r.a = t;
  for (int i = 0; i < NJ; ++i) r.v[i] = h * x.v[i];
  return r;
}

typedef struct {
  jet m[3][3];
} jmat3;

static jmat3 jmat_se2(jet x, jet y, jet th) { /* src/ceres_error.cpp:48-61 */
  jmat3 T;
  jet c = jcos(th), s = jsin(th);
  T.m[0][0] = c;
  T.m[0][1] = jneg(s);
  T.m[1][0] = s;
  T.m[1][1] = c;
  T.m[0][2] = x;
  T.m[1][2] = y;
  T.m[2][0] = jc(0.0);
  T.m[2][1] = jc(0.0);
  T.m[2][2] = jc(1.0);
  return T;
}
static jmat3 jmat_mul(const jmat3* A, const jmat3* B) {
  jmat3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      jet s = jmul(A->m[i][0], B->m[0][j]);
      s = jadd(s, jmul(A->m[i][1], B->m[1][j]));
      s = jadd(s, jmul(A->m[i][2], B->m[2][j]));
      C.m[i][j] = s;
    }
  return C;
}
/* generic 3x3 inverse by cofactors (what Eigen does for fixed size 3) */
static jmat3 jmat_inverse(const jmat3* A) {
  jmat3 R;
  jet cof[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i][j] = jsub(jmul(A->m[i1][j1], A->m[i2][j2]), jmul(A->m[i1][j2], A->m[i2][j1]));
    }
  jet det = jadd(jadd(jmul(A->m[0][0], cof[0][0]), jmul(A->m[0][1], cof[0][1])), jmul(A->m[0][2], cof[0][2]));
  jet invdet = jdiv(jc(1.0), det);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R.m[i][j] = jmul(cof[j][i], invdet);
  return R;
}

/* One residual block with Jacobian.  P1,P2: poses; m: (dx,dy,dtheta);
 * e[3]; J[18] = row-major 3x6 = [d e / d P1 | d e / d P2].                    */
/* Cholesky factor of the information matrix Omega = L L' (info6 = I11 I12 I13 I22 I23 I33, the reference's Edge
 * fields, include/graph.h:41-47).  Returns 0 when Omega is not positive definite.                              */
static int info_cholesky(const double* w, double L[6] /* L00 L10 L11 L20 L21 L22 */) {
  if (!(w[0] > 0.0)) return 0;
  L[0] = sqrt(w[0]);
  L[1] = w[1] / L[0];
  L[3] = w[2] / L[0];
  double d1 = w[3] - L[1] * L[1];
  if (!(d1 > 0.0)) return 0;
  L[2] = sqrt(d1);
  L[4] = (w[4] - L[3] * L[1]) / L[2];
  double d2 = w[5] - L[3] * L[3] - L[4] * L[4];
  if (!(d2 > 0.0)) return 0;
  L[5] = sqrt(d2);
  return 1;
}

/* info != NULL (optional mode, not on the reference's METHOD 0/1 path where the information entries are parsed but
 * unused): the residual is whitened, e_w = L' e_plain with Omega = L L', so that |e_w|^2 = e' Omega e -- the quantity
 * compute_edge_mahalanobis returns (src/layer_manager.cpp:230-282) -- and DCS takes its chi2 form
 * s = min(1, 2 phi / (phi + chi2)), e = s e_w (Agarwal et al. 2013; docs/code_Ex.png of the reference).          */
static void edge_functor_jet(const double* P1, const double* P2, const double* m, const double* info, int dcs,
                             double phi, double* e, double* J) {
  jmat3 wTa = jmat_se2(jvar(P1[0], 0), jvar(P1[1], 1), jvar(P1[2], 2));
  jmat3 wTb = jmat_se2(jvar(P2[0], 3), jvar(P2[1], 4), jvar(P2[2], 5));
  jmat3 aTb = jmat_se2(jc(m[0]), jc(m[1]), jc(m[2])); /* ctor :4-25 builds it in double, :80-83 casts */
  jmat3 ia = jmat_inverse(&wTa), im = jmat_inverse(&aTb);
  jmat3 t = jmat_mul(&ia, &wTb);
  jmat3 diff = jmat_mul(&im, &t); /* :87 */
  jet out[3];
  out[0] = diff.m[0][2];
  out[1] = diff.m[1][2];
  out[2] = jasin(diff.m[1][0]);
  if (info) {
    double L[6];
    if (!info_cholesky(info, L)) {
      for (int i = 0; i < 3; ++i) {
        e[i] = NAN;
        if (J)
          for (int k = 0; k < 6; ++k) J[6 * i + k] = NAN;
      }
      return;
    }
    jet w0 = jadd(jadd(jmul(jc(L[0]), out[0]), jmul(jc(L[1]), out[1])), jmul(jc(L[3]), out[2]));
    jet w1 = jadd(jmul(jc(L[2]), out[1]), jmul(jc(L[4]), out[2]));
    jet w2 = jmul(jc(L[5]), out[2]);
    out[0] = w0;
    out[1] = w1;
    out[2] = w2;
    if (dcs) {
      jet chi2 = jadd(jadd(jmul(w0, w0), jmul(w1, w1)), jmul(w2, w2));
      jet sc = jdiv(jmul(jc(2.0), jc(phi)), jadd(jc(phi), chi2));
      if (sc.a < 1.0) {
        out[0] = jmul(sc, out[0]);
        out[1] = jmul(sc, out[1]);
        out[2] = jmul(sc, out[2]);
      }
    }
  } else if (dcs) { /* :185-193 */
    jet res = jadd(jmul(diff.m[0][2], diff.m[0][2]), jmul(diff.m[1][2], diff.m[1][2]));
    jet psi_org = jsqrt(jdiv(jmul(jc(2.0), jc(phi)), jadd(jc(phi), res)));
    /* std::min(T(1.0), psi_org): returns psi_org iff psi_org < 1 (Jet compares scalar parts) */
    jet psi = (psi_org.a < 1.0) ? psi_org : jc(1.0);
    out[0] = jmul(psi, out[0]);
    out[1] = jmul(psi, out[1]);
    out[2] = jmul(psi, out[2]);
  }
  for (int i = 0; i < 3; ++i) {
    e[i] = out[i].a;
    if (J)
      for (int k = 0; k < 6; ++k) J[6 * i + k] = out[i].v[k];
  }
}

/* The same functor instantiated for T = double (Ceres' cost-only evaluations). */
static void mat_se2(double x, double y, double th, double T[3][3]) {
  double c = cos(th), s = sin(th);
  T[0][0] = c;
  T[0][1] = -s;
  T[0][2] = x;
  T[1][0] = s;
  T[1][1] = c;
  T[1][2] = y;
  T[2][0] = 0.0;
  T[2][1] = 0.0;
  T[2][2] = 1.0;
}
static void mat_inverse(double A[3][3], double R[3][3]) {
  double cof[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      cof[i][j] = A[i1][j1] * A[i2][j2] - A[i1][j2] * A[i2][j1];
    }
  double det = A[0][0] * cof[0][0] + A[0][1] * cof[0][1] + A[0][2] * cof[0][2];
  double invdet = 1.0 / det;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i][j] = cof[j][i] * invdet;
}
static void mat_mul(double A[3][3], double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
}
static void edge_functor_double(const double* P1, const double* P2, const double* m, const double* info, int dcs,
                                double phi, double* e) {
  double wTa[3][3], wTb[3][3], aTb[3][3], ia[3][3], im[3][3], t[3][3], diff[3][3];
  mat_se2(P1[0], P1[1], P1[2], wTa);
  mat_se2(P2[0], P2[1], P2[2], wTb);
  mat_se2(m[0], m[1], m[2], aTb);
  mat_inverse(wTa, ia);
  mat_inverse(aTb, im);
  mat_mul(ia, wTb, t);
  mat_mul(im, t, diff);
  e[0] = diff[0][2];
  e[1] = diff[1][2];
  e[2] = asin(diff[1][0]);
  if (info) {
    double L[6];
    if (!info_cholesky(info, L)) {
      e[0] = e[1] = e[2] = NAN;
      return;
    }
    double w0 = L[0] * e[0] + L[1] * e[1] + L[3] * e[2], w1 = L[2] * e[1] + L[4] * e[2], w2 = L[5] * e[2];
    e[0] = w0;
    e[1] = w1;
    e[2] = w2;
    if (dcs) {
      double chi2 = w0 * w0 + w1 * w1 + w2 * w2;
      double sc = 2.0 * phi / (phi + chi2);
      if (sc < 1.0) {
        e[0] *= sc;
        e[1] *= sc;
        e[2] *= sc;
      }
    }
  } else if (dcs) {
    double res = diff[0][2] * diff[0][2] + diff[1][2] * diff[1][2];
    double psi_org = sqrt(2.0 * phi / (phi + res));
    double psi = (psi_org < 1.0) ? psi_org : 1.0;
    e[0] *= psi;
    e[1] *= psi;
    e[2] *= psi;
  }
}

/* ceres::HuberLoss::Evaluate */
static void huber(double s, double a, double rho[3]) {
  double b = a * a;
  if (s > b) {
    double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = a / r;
    if (rho[1] < DBL_MIN) rho[1] = DBL_MIN;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s;
    rho[1] = 1.0;
    rho[2] = 0.0;
  }
}

/* ---------------------------------------------------------------- exports */
void pgo_oracle_edge(const double* P1, const double* P2, const double* meas, int dcs, double phi, double* e,
                     double* J /* 18 or NULL */) {
  if (J) edge_functor_jet(P1, P2, meas, NULL, dcs, phi, e, J);
  else edge_functor_double(P1, P2, meas, NULL, dcs, phi, e);
}
void pgo_oracle_edge_w(const double* P1, const double* P2, const double* meas, const double* info6, int dcs,
                       double phi, double* e, double* J /* 18 or NULL */) {
  if (J) edge_functor_jet(P1, P2, meas, info6, dcs, phi, e, J);
  else edge_functor_double(P1, P2, meas, info6, dcs, phi, e);
}

/* compute_edge_mahalanobis (src/layer_manager.cpp:230-282): m = r' Omega r of the PLAIN residual r = (ex, ey, etheta),
 * etheta = asin(clamp(diff[1][0], -1, 1)), m clamped at 0; any symmetric Omega (no factorisation).              */
void pgo_oracle_edge_chi2(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                          const double* info, double* out) {
  (void)N;
  for (int e = 0; e < E; ++e) {
    double wTa[3][3], wTb[3][3], aTb[3][3], iA[3][3], iM[3][3], t[3][3], diff[3][3];
    const double *P1 = poses + 3 * (size_t)ia[e], *P2 = poses + 3 * (size_t)ib[e], *m = meas + 3 * (size_t)e;
    const double* w = info + 6 * (size_t)e;
    mat_se2(P1[0], P1[1], P1[2], wTa);
    mat_se2(P2[0], P2[1], P2[2], wTb);
    mat_se2(m[0], m[1], m[2], aTb);
    mat_inverse(wTa, iA);
    mat_inverse(aTb, iM);
    mat_mul(iA, wTb, t);
    mat_mul(iM, t, diff);
    double ex = diff[0][2], ey = diff[1][2];
    double sd = diff[1][0] > 1.0 ? 1.0 : (diff[1][0] < -1.0 ? -1.0 : diff[1][0]);
    double et = asin(sd);
    double v = ex * (w[0] * ex + w[1] * ey + w[2] * et) + ey * (w[1] * ex + w[3] * ey + w[4] * et) +
               et * (w[2] * ex + w[4] * ey + w[5] * et);
    out[e] = v < 0.0 ? 0.0 : v;
  }
}

void pgo_oracle_huber(double s, double delta, double* rho3) { huber(s, delta, rho3); }

/* Evaluate every residual block.  kind[e] in {0,1,2}; DCS applies to kind != 0
 * when method == 1 (main.cpp:112-114,135-137).  apply_loss: scale r and J by
 * sqrt(rho') (Ceres Corrector with rho'' <= 0).  delta <= 0: no loss.
 * r: E x 3 or NULL, J: E x 18 or NULL.  Returns cost = 1/2 sum rho(|e|^2).
 * Returns NAN-poisoned cost when a residual/Jacobian entry is not finite.      */
double pgo_oracle_eval_w(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                         const double* info /* E x 6 or NULL */, const uint8_t* kind, int method, double phi,
                         double delta, int apply_loss, double* r, double* J, int threads) {
  (void)N;
  double cost = 0.0;
  int bad = 0;
#ifdef _OPENMP
  if (threads <= 0) threads = 1;
#pragma omp parallel for num_threads(threads) reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
  for (int e = 0; e < E; ++e) {
    double ee[3], JJ[18];
    int dcs = (method == 1 && kind[e] != 0);
    const double* P1 = poses + 3 * (size_t)ia[e];
    const double* P2 = poses + 3 * (size_t)ib[e];
    const double* w = info ? info + 6 * (size_t)e : NULL;
    if (J) edge_functor_jet(P1, P2, meas + 3 * (size_t)e, w, dcs, phi, ee, JJ);
    else edge_functor_double(P1, P2, meas + 3 * (size_t)e, w, dcs, phi, ee);
    double s = ee[0] * ee[0] + ee[1] * ee[1] + ee[2] * ee[2];
    double rho[3] = {s, 1.0, 0.0};
    if (delta > 0.0) huber(s, delta, rho);
    cost += 0.5 * rho[0];
    double sc = (apply_loss && delta > 0.0) ? sqrt(rho[1]) : 1.0;
    if (!isfinite(s)) bad |= 1;
    if (r)
      for (int k = 0; k < 3; ++k) r[3 * (size_t)e + k] = sc * ee[k];
    if (J)
      for (int k = 0; k < 18; ++k) {
        double v = sc * JJ[k];
        if (!isfinite(v)) bad |= 1;
        J[18 * (size_t)e + k] = v;
      }
  }
  return bad ? NAN : cost;
}
double pgo_oracle_eval(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                       const uint8_t* kind, int method, double phi, double delta, int apply_loss, double* r,
                       double* J, int threads) {
  return pgo_oracle_eval_w(N, poses, E, ia, ib, meas, NULL, kind, method, phi, delta, apply_loss, r, J, threads);
}

/* METHOD 2, switchable constraints (main.cpp:115-125,138-145; src/ceres_error.cpp:237-317).  Residual blocks:
 *   odometry edges            OdometryResidue, Huber
 *   closure + bogus edges     SwitchableClosureResidue  e = s * e_plain(P1,P2)  (3 residuals; params P1, P2, s), Huber
 *                             SwitchPriorResidue        sqrt(lambda) * (1 - s)  (1 residual; param s), no loss
 * sw[e] is the switch of edge e (ignored for odometry edges).  Outputs (any may be NULL): r E x 3, J E x 18 (w.r.t. the
 * poses), Js E x 3 (w.r.t. the switch; 0 for odometry), q E (prior residual; 0 for odometry).  The Huber corrector scales
 * r, J and Js of a block by the same sqrt(rho').  Returns cost = 1/2 sum rho(|e|^2) + 1/2 sum q^2.                     */
double pgo_oracle_eval_sc(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                          const uint8_t* kind, const double* sw, double lambda, double delta, int apply_loss, double* r,
                          double* J, double* Js, double* q, int threads) {
  (void)N;
  double cost = 0.0;
  int bad = 0;
  const double sl = sqrt(lambda);
#ifdef _OPENMP
  if (threads <= 0) threads = 1;
#pragma omp parallel for num_threads(threads) reduction(+ : cost) reduction(| : bad) schedule(static)
#endif
  for (int e = 0; e < E; ++e) {
    double ep[3], Jp[18];
    const double* P1 = poses + 3 * (size_t)ia[e];
    const double* P2 = poses + 3 * (size_t)ib[e];
    const int want_j = (J != NULL) || (Js != NULL);
    if (want_j) edge_functor_jet(P1, P2, meas + 3 * (size_t)e, NULL, 0, 0.5, ep, Jp);
    else edge_functor_double(P1, P2, meas + 3 * (size_t)e, NULL, 0, 0.5, ep);
    const int sc_edge = kind[e] != 0;
    const double sv = sc_edge ? sw[e] : 1.0;
    double ee[3] = {sv * ep[0], sv * ep[1], sv * ep[2]};  /* product rule of Jet<7>: d e / d P = s d e_p / d P, d e / d s = e_p */
    double s2 = ee[0] * ee[0] + ee[1] * ee[1] + ee[2] * ee[2];
    double rho[3] = {s2, 1.0, 0.0};
    if (delta > 0.0) huber(s2, delta, rho);
    cost += 0.5 * rho[0];
    const double scl = (apply_loss && delta > 0.0) ? sqrt(rho[1]) : 1.0;
    if (!isfinite(s2)) bad |= 1;
    double prior = 0.0;
    if (sc_edge) {
      prior = sl * (1.0 - sv);
      cost += 0.5 * prior * prior;
    }
    if (q) q[e] = prior;
    if (r)
      for (int k = 0; k < 3; ++k) r[3 * (size_t)e + k] = scl * ee[k];
    if (J)
      for (int k = 0; k < 18; ++k) {
        double v = scl * sv * Jp[k];
        if (!isfinite(v)) bad |= 1;
        J[18 * (size_t)e + k] = v;
      }
    if (Js)
      for (int k = 0; k < 3; ++k) Js[3 * (size_t)e + k] = sc_edge ? scl * ep[k] : 0.0;
  }
  return bad ? NAN : cost;
}

/* =====================================================================
 * LM + block-Jacobi PCG: the "port" CPU baseline (same algorithm as the HIP
 * backend, plain C + OpenMP).  LM policy = Ceres TrustRegionMinimizer +
 * LevenbergMarquardtStrategy defaults.
 * ===================================================================== */
typedef struct {
  int32_t method, max_iters, fixed_pose, jacobi_scaling;
  double phi, huber_delta, ftol, gtol, ptol, radius0, max_radius, min_radius, min_relative_decrease, min_lm_diagonal,
      max_lm_diagonal, pcg_rtol;
  int32_t pcg_max_iters, threads, verbose, block_poses; /* block_poses: poses per block-Jacobi block (<=1: 3x3) */
  int32_t chain_len, _pad; /* > 0: block-Jacobi over segments of chain_len consecutive poses whose blocks are kept
                              block-TRIDIAGONAL (the odometry chain inside the segment; every other edge only adds to
                              the 3x3 diagonal blocks), solved exactly by a block LDL' sweep; overrides block_poses */
} oracle_options;

typedef struct {
  int32_t iter, step_ok;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius;
  int32_t pcg_iters, _pad;
  double pcg_rel_residual, seconds;
} oracle_iter;

typedef struct {
  int32_t termination, iterations, successful_steps, total_pcg_iters;
  double initial_cost, final_cost, seconds_total, seconds_eval, seconds_assemble, seconds_linear, seconds_candidate;
} oracle_summary;

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
  int N, E;
  const int32_t *ia, *ib;
  int32_t* inc_ptr;  /* N+1 */
  int32_t* inc_edge; /* (e<<1)|side */
  int32_t* inc_col;
  double* Hd;   /* N x 9 */
  double* Hoff; /* n_inc x 9: (J_self * S_self)^T (J_other * S_other) */
  double* Minv; /* N x 9 */
  double* D2;   /* 3N */
  int B;        /* poses per preconditioner block; > 1: dense Cholesky factors in Lg */
  double* Lg;   /* n_groups x nb x nb (lower Cholesky factors), nb = 3B */
  int chain;    /* > 0: chain preconditioner with segments of this many poses */
  double* Cw;   /* N x 9: W_i = C_i S_{i-1}^-1 (C_i = the block (i, i-1) of H; 0 at a segment start) */
  double* Cs;   /* N x 9: S_i^-1,  S_i = (H_ii + D2_i) - W_i C_i' */
} normal_eq;

static void build_incidence(normal_eq* Q) {
  int N = Q->N, E = Q->E;
  Q->inc_ptr = (int32_t*)calloc((size_t)N + 1, sizeof(int32_t));
  for (int e = 0; e < E; ++e) {
    Q->inc_ptr[Q->ia[e] + 1]++;
    Q->inc_ptr[Q->ib[e] + 1]++;
  }
  for (int i = 0; i < N; ++i) Q->inc_ptr[i + 1] += Q->inc_ptr[i];
  size_t n_inc = (size_t)Q->inc_ptr[N];
  Q->inc_edge = (int32_t*)malloc(n_inc * sizeof(int32_t));
  Q->inc_col = (int32_t*)malloc(n_inc * sizeof(int32_t));
  int32_t* fill = (int32_t*)malloc((size_t)N * sizeof(int32_t));
  memcpy(fill, Q->inc_ptr, (size_t)N * sizeof(int32_t));
  for (int e = 0; e < E; ++e) {
    int a = Q->ia[e], b = Q->ib[e];
    int q = fill[a]++;
    Q->inc_edge[q] = (e << 1);
    Q->inc_col[q] = b;
    q = fill[b]++;
    Q->inc_edge[q] = (e << 1) | 1;
    Q->inc_col[q] = a;
  }
  free(fill);
  Q->Hd = (double*)malloc((size_t)N * 9 * sizeof(double));
  Q->Hoff = (double*)malloc(n_inc * 9 * sizeof(double));
  Q->Minv = (double*)malloc((size_t)N * 9 * sizeof(double));
  Q->D2 = (double*)malloc((size_t)N * 3 * sizeof(double));
}
static void free_normal_eq(normal_eq* Q) {
  free(Q->inc_ptr);
  free(Q->inc_edge);
  free(Q->inc_col);
  free(Q->Hd);
  free(Q->Hoff);
  free(Q->Minv);
  free(Q->D2);
  free(Q->Lg);
  free(Q->Cw);
  free(Q->Cs);
}

/* H = (J S)^T (J S), gs = S J^T r ; s: 3N column scales (0 on the fixed pose) */
static void assemble(normal_eq* Q, const double* r, const double* J, const double* s, double* gs, int threads) {
  int N = Q->N;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int i = 0; i < N; ++i) {
    double hd[9] = {0}, g[3] = {0};
    const double* si = s + 3 * (size_t)i;
    for (int q = Q->inc_ptr[i]; q < Q->inc_ptr[i + 1]; ++q) {
      int e = Q->inc_edge[q] >> 1, side = Q->inc_edge[q] & 1;
      const double* Je = J + 18 * (size_t)e;
      const double* re = r + 3 * (size_t)e;
      const double* so = s + 3 * (size_t)Q->inc_col[q];
      double* ho = Q->Hoff + 9 * (size_t)q;
      int cs = side ? 3 : 0, co = side ? 0 : 3;
      for (int a = 0; a < 3; ++a) {
        double ga = 0.0;
        for (int k = 0; k < 3; ++k) ga += Je[6 * k + cs + a] * re[k];
        g[a] += si[a] * ga;
        for (int b = 0; b < 3; ++b) {
          double dd = 0.0, oo = 0.0;
          for (int k = 0; k < 3; ++k) {
            dd += Je[6 * k + cs + a] * Je[6 * k + cs + b];
            oo += Je[6 * k + cs + a] * Je[6 * k + co + b];
          }
          hd[3 * a + b] += si[a] * si[b] * dd;
          ho[3 * a + b] = si[a] * so[b] * oo;
        }
      }
    }
    memcpy(Q->Hd + 9 * (size_t)i, hd, sizeof hd);
    gs[3 * (size_t)i + 0] = g[0];
    gs[3 * (size_t)i + 1] = g[1];
    gs[3 * (size_t)i + 2] = g[2];
  }
}

static void inv3_sym(const double* A, double* R) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  R[0] = c00 * id;
  R[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  R[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  R[3] = c01 * id;
  R[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  R[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  R[6] = c02 * id;
  R[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  R[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

/* dense Cholesky factors of the diagonal blocks of (H + D2) over groups of B consecutive poses */
static void factor_groups(normal_eq* Q, int threads) {
  int N = Q->N, B = Q->B, nb = 3 * B, ng = (N + B - 1) / B;
  if (!Q->Lg) Q->Lg = (double*)malloc((size_t)ng * nb * nb * sizeof(double));
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int g = 0; g < ng; ++g) {
    double* M = Q->Lg + (size_t)g * nb * nb;
    int g0 = g * B, g1 = g0 + B < N ? g0 + B : N;
    memset(M, 0, (size_t)nb * nb * sizeof(double));
    for (int i = 3 * (g1 - g0); i < nb; ++i) M[i * nb + i] = 1.0;
    for (int row = g0; row < g1; ++row) {
      int r = 3 * (row - g0);
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) M[(r + a) * nb + r + b] = Q->Hd[9 * (size_t)row + 3 * a + b];
        M[(r + a) * nb + r + a] += Q->D2[3 * (size_t)row + a];
      }
      for (int q = Q->inc_ptr[row]; q < Q->inc_ptr[row + 1]; ++q) {
        int col = Q->inc_col[q];
        if (col < g0 || col >= g1) continue;
        const double* ho = Q->Hoff + 9 * (size_t)q;
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) M[(r + a) * nb + 3 * (col - g0) + b] += ho[3 * a + b];
      }
    }
    for (int j = 0; j < nb; ++j) { /* in-place lower Cholesky */
      double d = M[j * nb + j];
      for (int k = 0; k < j; ++k) d -= M[j * nb + k] * M[j * nb + k];
      d = sqrt(d);
      M[j * nb + j] = d;
      for (int i = j + 1; i < nb; ++i) {
        double v = M[i * nb + j];
        for (int k = 0; k < j; ++k) v -= M[i * nb + k] * M[j * nb + k];
        M[i * nb + j] = v / d;
      }
    }
  }
}

static void apply_groups(const normal_eq* Q, const double* r, double* z, int threads) {
  int N = Q->N, B = Q->B, nb = 3 * B, ng = (N + B - 1) / B;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int g = 0; g < ng; ++g) {
    const double* L = Q->Lg + (size_t)g * nb * nb;
    int g0 = g * B, g1 = g0 + B < N ? g0 + B : N, m = 3 * (g1 - g0);
    double w[96];
    for (int i = 0; i < m; ++i) {
      double v = r[3 * (size_t)g0 + i];
      for (int k = 0; k < i; ++k) v -= L[i * nb + k] * w[k];
      w[i] = v / L[i * nb + i];
    }
    for (int i = m - 1; i >= 0; --i) {
      double v = w[i];
      for (int k = i + 1; k < m; ++k) v -= L[k * nb + i] * w[k];
      w[i] = v / L[i * nb + i];
    }
    for (int i = 0; i < m; ++i) z[3 * (size_t)g0 + i] = w[i];
  }
}

/* Chain preconditioner: M = block-tridiagonal part of (H + D2) inside segments of Q->chain consecutive poses
 * (= sum of J'J over the edges joining consecutive poses of a segment + the 3x3 diagonal blocks of every other edge
 * + D2, hence SPD).  Block LDL': S_i = M_ii - W_i C_i', W_i = C_i S_{i-1}^-1.                                      */
static void factor_chain(normal_eq* Q, int threads) {
  int N = Q->N, L = Q->chain, ns = (N + L - 1) / L;
  if (!Q->Cw) Q->Cw = (double*)malloc((size_t)N * 9 * sizeof(double));
  if (!Q->Cs) Q->Cs = (double*)malloc((size_t)N * 9 * sizeof(double));
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int sgm = 0; sgm < ns; ++sgm) {
    int s0 = sgm * L, s1 = s0 + L < N ? s0 + L : N;
    for (int i = s0; i < s1; ++i) {
      double M[9], C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double* W = Q->Cw + 9 * (size_t)i;
      for (int k = 0; k < 9; ++k) M[k] = Q->Hd[9 * (size_t)i + k];
      for (int a = 0; a < 3; ++a) M[4 * a] += Q->D2[3 * (size_t)i + a];
      if (i > s0) {
        for (int q = Q->inc_ptr[i]; q < Q->inc_ptr[i + 1]; ++q)
          if (Q->inc_col[q] == i - 1)
            for (int k = 0; k < 9; ++k) C[k] += Q->Hoff[9 * (size_t)q + k];
        const double* Sp = Q->Cs + 9 * (size_t)(i - 1);
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) W[3 * a + b] = C[3 * a] * Sp[b] + C[3 * a + 1] * Sp[3 + b] + C[3 * a + 2] * Sp[6 + b];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) M[3 * a + b] -= W[3 * a] * C[3 * b] + W[3 * a + 1] * C[3 * b + 1] + W[3 * a + 2] * C[3 * b + 2];
      } else {
        for (int k = 0; k < 9; ++k) W[k] = 0.0;
      }
      /* S is symmetric in exact arithmetic: keep the upper triangle */
      M[3] = M[1];
      M[6] = M[2];
      M[7] = M[5];
      inv3_sym(M, Q->Cs + 9 * (size_t)i);
    }
  }
}

static void apply_chain(const normal_eq* Q, const double* r, double* z, int threads) {
  int N = Q->N, L = Q->chain, ns = (N + L - 1) / L;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int sgm = 0; sgm < ns; ++sgm) {
    int s0 = sgm * L, s1 = s0 + L < N ? s0 + L : N;
    double t[3] = {0, 0, 0};
    for (int i = s0; i < s1; ++i) { /* t_i = r_i - W_i t_{i-1};  z_i <- S_i^-1 t_i */
      const double *W = Q->Cw + 9 * (size_t)i, *Si = Q->Cs + 9 * (size_t)i;
      double v[3];
      for (int a = 0; a < 3; ++a) v[a] = r[3 * (size_t)i + a] - (W[3 * a] * t[0] + W[3 * a + 1] * t[1] + W[3 * a + 2] * t[2]);
      for (int a = 0; a < 3; ++a) {
        t[a] = v[a];
      }
      for (int a = 0; a < 3; ++a) z[3 * (size_t)i + a] = Si[3 * a] * v[0] + Si[3 * a + 1] * v[1] + Si[3 * a + 2] * v[2];
    }
    for (int i = s1 - 2; i >= s0; --i) { /* z_i -= W_{i+1}' z_{i+1} */
      const double* W = Q->Cw + 9 * (size_t)(i + 1);
      const double* zn = z + 3 * (size_t)(i + 1);
      for (int a = 0; a < 3; ++a) z[3 * (size_t)i + a] -= W[a] * zn[0] + W[3 + a] * zn[1] + W[6 + a] * zn[2];
    }
  }
}

/* y = (H + D2) x */
static void spmv(const normal_eq* Q, const double* x, double* y, int with_d2, int threads) {
  int N = Q->N;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int i = 0; i < N; ++i) {
    const double* h = Q->Hd + 9 * (size_t)i;
    const double* xi = x + 3 * (size_t)i;
    double acc[3];
    for (int a = 0; a < 3; ++a) {
      acc[a] = h[3 * a] * xi[0] + h[3 * a + 1] * xi[1] + h[3 * a + 2] * xi[2];
      if (with_d2) acc[a] += Q->D2[3 * (size_t)i + a] * xi[a];
    }
    for (int q = Q->inc_ptr[i]; q < Q->inc_ptr[i + 1]; ++q) {
      const double* ho = Q->Hoff + 9 * (size_t)q;
      const double* xc = x + 3 * (size_t)Q->inc_col[q];
      for (int a = 0; a < 3; ++a) acc[a] += ho[3 * a] * xc[0] + ho[3 * a + 1] * xc[1] + ho[3 * a + 2] * xc[2];
    }
    y[3 * (size_t)i] = acc[0];
    y[3 * (size_t)i + 1] = acc[1];
    y[3 * (size_t)i + 2] = acc[2];
  }
}

static double dot(const double* a, const double* b, size_t n, int threads) {
  double s = 0.0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) reduction(+ : s) schedule(static)
#endif
  for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

static void apply_minv(const normal_eq* Q, const double* r, double* z, int threads) {
  int N = Q->N;
  if (Q->chain > 0) {
    apply_chain(Q, r, z, threads);
    return;
  }
  if (Q->B > 1) {
    apply_groups(Q, r, z, threads);
    return;
  }
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int i = 0; i < N; ++i) {
    const double* m = Q->Minv + 9 * (size_t)i;
    const double* ri = r + 3 * (size_t)i;
    for (int a = 0; a < 3; ++a) z[3 * (size_t)i + a] = m[3 * a] * ri[0] + m[3 * a + 1] * ri[1] + m[3 * a + 2] * ri[2];
  }
}

/* block-Jacobi PCG on (H + D2) y = b from y = 0.  Returns iterations. */
static int pcg(const normal_eq* Q, const double* b, double* y, double rtol, int max_iters, double* rel_out,
               double* w /* 4 x 3N workspace */, int threads) {
  size_t n = (size_t)3 * Q->N;
  double *r = w, *z = w + n, *p = w + 2 * n, *Ap = w + 3 * n;
  memset(y, 0, n * sizeof(double));
  memcpy(r, b, n * sizeof(double));
  double bnorm = sqrt(dot(b, b, n, threads));
  if (bnorm == 0.0) {
    *rel_out = 0.0;
    return 0;
  }
  apply_minv(Q, r, z, threads);
  memcpy(p, z, n * sizeof(double));
  double rz = dot(r, z, n, threads);
  double rel = 1.0;
  int it = 0;
  while (it < max_iters) {
    spmv(Q, p, Ap, 1, threads);
    double pAp = dot(p, Ap, n, threads);
    double alpha = rz / pAp;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (size_t i = 0; i < n; ++i) {
      y[i] += alpha * p[i];
      r[i] -= alpha * Ap[i];
    }
    ++it;
    rel = sqrt(dot(r, r, n, threads)) / bnorm;
    if (rel <= rtol) break;
    apply_minv(Q, r, z, threads);
    double rz_new = dot(r, z, n, threads);
    double beta = rz_new / rz;
    rz = rz_new;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (size_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
  }
  *rel_out = rel;
  return it;
}

int pgo_oracle_lm_pcg_w(int N, double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                        const double* info /* E x 6 or NULL */, const uint8_t* kind, const oracle_options* o,
                        oracle_iter* recs, int cap, int* n_recs, oracle_summary* sum) {
  int threads = o->threads > 0 ? o->threads : 1;
  size_t n = (size_t)3 * N;
  normal_eq Q;
  memset(&Q, 0, sizeof Q);
  Q.N = N;
  Q.E = E;
  Q.ia = ia;
  Q.ib = ib;
  Q.B = (o->block_poses > 1) ? (o->block_poses > 32 ? 32 : o->block_poses) : 1;
  Q.chain = o->chain_len > 0 ? o->chain_len : 0;
  build_incidence(&Q);
  double* r = (double*)malloc((size_t)E * 3 * sizeof(double));
  double* J = (double*)malloc((size_t)E * 18 * sizeof(double));
  double* s = (double*)malloc(n * sizeof(double));
  double* gs = (double*)malloc(n * sizeof(double));
  double* y = (double*)malloc(n * sizeof(double));
  double* cand = (double*)malloc(n * sizeof(double));
  double* w = (double*)malloc(5 * n * sizeof(double));
  double* Hy = w + 4 * n;
  int nrec = 0;
  double t_start = now_s(), t_eval = 0, t_asm = 0, t_lin = 0, t_cand = 0;
  int term = 4, iters = 0, succ = 0, tot_pcg = 0;
  int fixed = o->fixed_pose;

  double t0 = now_s();
  double cost = pgo_oracle_eval_w(N, poses, E, ia, ib, meas, info, kind, o->method, o->phi, o->huber_delta, 1, r, J, threads);
  t_eval += now_s() - t0;
  double initial_cost = cost;
  if (!isfinite(cost)) {
    term = 6;
    goto done;
  }
  /* Jacobi scaling from the iteration-0 Jacobian: 1 / (1 + ||col||) */
  t0 = now_s();
  for (size_t i = 0; i < n; ++i) s[i] = 1.0;
  if (fixed >= 0) s[3 * fixed] = s[3 * fixed + 1] = s[3 * fixed + 2] = 0.0;
  assemble(&Q, r, J, s, gs, threads);
  if (o->jacobi_scaling) {
    for (int i = 0; i < N; ++i)
      for (int a = 0; a < 3; ++a) s[3 * (size_t)i + a] = 1.0 / (1.0 + sqrt(Q.Hd[9 * (size_t)i + 4 * a]));
    if (fixed >= 0) s[3 * fixed] = s[3 * fixed + 1] = s[3 * fixed + 2] = 0.0;
    assemble(&Q, r, J, s, gs, threads);
  }
  t_asm += now_s() - t0;
  double gmax = 0.0;
  for (size_t i = 0; i < n; ++i)
    if (s[i] > 0.0) {
      double g = fabs(gs[i] / s[i]);
      if (g > gmax) gmax = g;
    }
  double x_norm = 0.0;
  for (int i = 0; i < N; ++i)
    if (i != fixed)
      for (int a = 0; a < 3; ++a) x_norm += poses[3 * (size_t)i + a] * poses[3 * (size_t)i + a];
  x_norm = sqrt(x_norm);
  double radius = o->radius0, decrease_factor = 2.0;
  int prev_success = 1, invalid_run = 0;
  if (nrec < cap) {
    oracle_iter R;
    memset(&R, 0, sizeof R);
    R.iter = 0;
    R.step_ok = 1;
    R.cost = cost;
    R.gradient_max_norm = gmax;
    R.radius = radius;
    R.seconds = now_s() - t_start;
    recs[nrec++] = R;
  }
  if (o->verbose) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  pcg_iter\n");
  if (o->verbose) printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e  %d\n", 0, cost, 0.0, gmax, 0.0, 0.0, radius, 0);

  for (int iter = 1;; ++iter) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iter > o->max_iters) {
      term = 4;
      break;
    }
    if (prev_success && gmax <= o->gtol) {
      term = 2;
      break;
    }
    if (radius < o->min_radius) {
      term = 5;
      break;
    }
    double it0 = now_s();
    iters = iter;
    /* LM diagonal from the (scaled) Jacobian's squared column norms */
    for (int i = 0; i < N; ++i) {
      double A[9];
      memcpy(A, Q.Hd + 9 * (size_t)i, sizeof A);
      for (int a = 0; a < 3; ++a) {
        double d = A[4 * a];
        if (d < o->min_lm_diagonal) d = o->min_lm_diagonal;
        if (d > o->max_lm_diagonal) d = o->max_lm_diagonal;
        double d2 = d / radius;
        if (i == fixed) d2 = 1.0;
        Q.D2[3 * (size_t)i + a] = d2;
        A[4 * a] += d2;
      }
      inv3_sym(A, Q.Minv + 9 * (size_t)i);
    }
    if (Q.chain > 0) factor_chain(&Q, threads);
    else if (Q.B > 1) factor_groups(&Q, threads);
    t0 = now_s();
    double rel = 0.0;
    int k = pcg(&Q, gs, y, o->pcg_rtol, o->pcg_max_iters, &rel, w, threads);
    tot_pcg += k;
    /* model_cost_change = -(J d).(r + J d / 2) with d = -S y  ==  y.gs - y.(H y)/2 */
    spmv(&Q, y, Hy, 0, threads);
    double model = dot(y, gs, n, threads) - 0.5 * dot(y, Hy, n, threads);
    t_lin += now_s() - t0;
    oracle_iter R;
    memset(&R, 0, sizeof R);
    R.iter = iter;
    R.pcg_iters = k;
    R.pcg_rel_residual = rel;
    int finite_step = 1;
    for (size_t i = 0; i < n; ++i)
      if (!isfinite(y[i])) finite_step = 0;
    if (!finite_step || !(model > 0.0)) { /* invalid step */
      if (++invalid_run >= 5) {
        term = 6;
        break;
      }
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      prev_success = 0;
      R.step_ok = -1;
      R.cost = cost;
      R.radius = radius;
      R.gradient_max_norm = gmax;
      R.seconds = now_s() - it0;
      if (nrec < cap) recs[nrec++] = R;
      continue;
    }
    invalid_run = 0;
    double step2 = 0.0;
    for (size_t i = 0; i < n; ++i) {
      double d = -s[i] * y[i];
      cand[i] = poses[i] + d;
      step2 += d * d;
    }
    t0 = now_s();
    double cand_cost =
        pgo_oracle_eval_w(N, cand, E, ia, ib, meas, info, kind, o->method, o->phi, o->huber_delta, 1, NULL, NULL, threads);
    t_cand += now_s() - t0;
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;
    R.step_norm = sqrt(step2);
    R.cost_change = cost - cand_cost;
    R.gradient_max_norm = gmax;
    if (R.step_norm <= o->ptol * (x_norm + o->ptol)) {
      term = 3;
      R.cost = cost;
      R.radius = radius;
      R.seconds = now_s() - it0;
      if (nrec < cap) recs[nrec++] = R;
      break;
    }
    if (fabs(R.cost_change) <= o->ftol * cost) {
      term = 1;
      R.cost = cost;
      R.radius = radius;
      R.seconds = now_s() - it0;
      if (nrec < cap) recs[nrec++] = R;
      break;
    }
    double rho = (cand_cost >= DBL_MAX) ? -DBL_MAX : R.cost_change / model;
    R.relative_decrease = rho;
    if (rho > o->min_relative_decrease) {
      memcpy(poses, cand, n * sizeof(double));
      x_norm = 0.0;
      for (int i = 0; i < N; ++i)
        if (i != fixed)
          for (int a = 0; a < 3; ++a) x_norm += poses[3 * (size_t)i + a] * poses[3 * (size_t)i + a];
      x_norm = sqrt(x_norm);
      t0 = now_s();
      cost = pgo_oracle_eval_w(N, poses, E, ia, ib, meas, info, kind, o->method, o->phi, o->huber_delta, 1, r, J, threads);
      t_eval += now_s() - t0;
      if (!isfinite(cost)) {
        term = 6;
        break;
      }
      t0 = now_s();
      assemble(&Q, r, J, s, gs, threads);
      t_asm += now_s() - t0;
      gmax = 0.0;
      for (size_t i = 0; i < n; ++i)
        if (s[i] > 0.0) {
          double g = fabs(gs[i] / s[i]);
          if (g > gmax) gmax = g;
        }
      double t = 2.0 * rho - 1.0;
      double f = 1.0 - t * t * t;
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f;
      if (radius > o->max_radius) radius = o->max_radius;
      decrease_factor = 2.0;
      prev_success = 1;
      ++succ;
      R.step_ok = 1;
      R.cost = cost;
      R.gradient_max_norm = gmax;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      prev_success = 0;
      R.step_ok = 0;
      R.cost = cand_cost;
    }
    R.radius = radius;
    R.seconds = now_s() - it0;
    if (nrec < cap) recs[nrec++] = R;
    if (o->verbose)
      printf("%4d % .6e  % .2e  % .2e  % .2e  % .2e  % .2e  %d\n", iter, R.cost, R.cost_change, gmax, R.step_norm, rho,
             radius, k);
  }
done:
  if (sum) {
    sum->termination = term;
    sum->iterations = iters;
    sum->successful_steps = succ;
    sum->total_pcg_iters = tot_pcg;
    sum->initial_cost = initial_cost;
    sum->final_cost = cost;
    sum->seconds_total = now_s() - t_start;
    sum->seconds_eval = t_eval;
    sum->seconds_assemble = t_asm;
    sum->seconds_linear = t_lin;
    sum->seconds_candidate = t_cand;
  }
  if (n_recs) *n_recs = nrec;
  free(r);
  free(J);
  free(s);
  free(gs);
  free(y);
  free(cand);
  free(w);
  free_normal_eq(&Q);
  return 0;
}

/* y = H x (+ D2 x) for an explicit scale vector / radius: SpMV parity helper.
 * Builds the scaled normal equations at `poses` and multiplies.               */
int pgo_oracle_normal_eq_w(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                           const double* info /* E x 6 or NULL */, const uint8_t* kind, int method, double phi,
                           double delta, int fixed_pose,
                         const double* s_or_null, double* g_out /*3N*/, double* hdiag_out /*N x 9*/,
                         const double* x_or_null, double* y_or_null, int threads) {
  size_t n = (size_t)3 * N;
  normal_eq Q;
  memset(&Q, 0, sizeof Q);
  Q.N = N;
  Q.E = E;
  Q.ia = ia;
  Q.ib = ib;
  build_incidence(&Q);
  double* r = (double*)malloc((size_t)E * 3 * sizeof(double));
  double* J = (double*)malloc((size_t)E * 18 * sizeof(double));
  double* s = (double*)malloc(n * sizeof(double));
  double* gs = (double*)malloc(n * sizeof(double));
  pgo_oracle_eval_w(N, poses, E, ia, ib, meas, info, kind, method, phi, delta, 1, r, J, threads);
  for (size_t i = 0; i < n; ++i) s[i] = s_or_null ? s_or_null[i] : 1.0;
  if (fixed_pose >= 0) s[3 * fixed_pose] = s[3 * fixed_pose + 1] = s[3 * fixed_pose + 2] = 0.0;
  assemble(&Q, r, J, s, gs, threads);
  if (g_out) memcpy(g_out, gs, n * sizeof(double));
  if (hdiag_out) memcpy(hdiag_out, Q.Hd, (size_t)N * 9 * sizeof(double));
  if (x_or_null && y_or_null) spmv(&Q, x_or_null, y_or_null, 0, threads);
  free(r);
  free(J);
  free(s);
  free(gs);
  free_normal_eq(&Q);
  return 0;
}

int pgo_oracle_lm_pcg(int N, double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                      const uint8_t* kind, const oracle_options* o, oracle_iter* recs, int cap, int* n_recs,
                      oracle_summary* sum) {
  return pgo_oracle_lm_pcg_w(N, poses, E, ia, ib, meas, NULL, kind, o, recs, cap, n_recs, sum);
}
int pgo_oracle_normal_eq(int N, const double* poses, int E, const int32_t* ia, const int32_t* ib, const double* meas,
                         const uint8_t* kind, int method, double phi, double delta, int fixed_pose,
                         const double* s_or_null, double* g_out, double* hdiag_out, const double* x_or_null,
                         double* y_or_null, int threads) {
  return pgo_oracle_normal_eq_w(N, poses, E, ia, ib, meas, NULL, kind, method, phi, delta, fixed_pose, s_or_null, g_out,
                                hdiag_out, x_or_null, y_or_null, threads);
}
