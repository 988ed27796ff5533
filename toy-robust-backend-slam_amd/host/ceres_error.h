// Residual functors with the reference's signature (DCS-ceres/include/ceres_error.h:9-36):
//     X(dx, dy, dtheta);  template<T> bool operator()(const T* P1, const T* P2, T* e) const;  static Create(...)
// On this backend the functor is a *descriptor*: Create() returns the edge description that
// pgo::Problem hands to the HIP kernel (which evaluates residual, Jacobian, DCS weight and Huber scaling for all
// edges at once); operator() stays host-callable (T = double or any arithmetic type with sin/cos/asin/sqrt
// overloads) for unit tests and spot checks -- closed form of src/ceres_error.cpp:42-94 / :135-196.
#ifndef PGO_HOST_CERES_ERROR_H_
#define PGO_HOST_CERES_ERROR_H_

#include <cmath>

namespace pgo {
struct CostFunction {   // stands in for ceres::CostFunction* on this path
  enum Kind { PLAIN, DCS, SWITCHABLE, SWITCH_PRIOR };
  double dx, dy, dtheta;   // SWITCH_PRIOR: dx holds lambda
  Kind kind;
};
}  // namespace pgo

namespace pgo_detail {
template <typename T>
inline void se2_error(const T* P1, const T* P2, double dx, double dy, double dtheta, T* e) {
  using std::asin; using std::cos; using std::sin;
  const T c1 = cos(P1[2]), s1 = sin(P1[2]);
  const T Dx = P2[0] - P1[0], Dy = P2[1] - P1[1];
  const T a = c1 * Dx + s1 * Dy - T(dx), b = -s1 * Dx + c1 * Dy - T(dy);
  const double cd = std::cos(dtheta), sd = std::sin(dtheta);
  e[0] = T(cd) * a + T(sd) * b;
  e[1] = -T(sd) * a + T(cd) * b;
  e[2] = asin(sin(P2[2] - P1[2] - T(dtheta)));
}
}  // namespace pgo_detail

struct OdometryResidue {
  OdometryResidue(double dx_, double dy_, double dtheta_) : dx(dx_), dy(dy_), dtheta(dtheta_) {}
  template <typename T>
  bool operator()(const T* const P1, const T* const P2, T* e) const {
    pgo_detail::se2_error(P1, P2, dx, dy, dtheta, e);
    return true;
  }
  static pgo::CostFunction* Create(double dx, double dy, double dtheta) { return new pgo::CostFunction{dx, dy, dtheta, pgo::CostFunction::PLAIN}; }
  double dx, dy, dtheta;
};

struct DCSClosureResidue {
  DCSClosureResidue(double dx_, double dy_, double dtheta_) : dx(dx_), dy(dy_), dtheta(dtheta_) {}
  template <typename T>
  bool operator()(const T* const P1, const T* const P2, T* e) const {
    using std::sqrt;
    pgo_detail::se2_error(P1, P2, dx, dy, dtheta, e);
    const double phi = 0.5;  // src/ceres_error.cpp:185
    const T res = e[0] * e[0] + e[1] * e[1];
    const T psi_org = sqrt(T(2.0 * phi) / (T(phi) + res));
    if (psi_org < T(1.0)) {
      e[0] = psi_org * e[0];
      e[1] = psi_org * e[1];
      e[2] = psi_org * e[2];
    }
    return true;
  }
  static pgo::CostFunction* Create(double dx, double dy, double dtheta) { return new pgo::CostFunction{dx, dy, dtheta, pgo::CostFunction::DCS}; }
  double dx, dy, dtheta;
};

// Switchable constraints (reference include/ceres_error.h:38-66, src/ceres_error.cpp:199-317)
struct SwitchableClosureResidue {
  SwitchableClosureResidue(double dx_, double dy_, double dtheta_) : dx(dx_), dy(dy_), dtheta(dtheta_) {}
  template <typename T>
  bool operator()(const T* const P1, const T* const P2, const T* const S, T* e) const {
    pgo_detail::se2_error(P1, P2, dx, dy, dtheta, e);
    e[0] = S[0] * e[0];
    e[1] = S[0] * e[1];
    e[2] = S[0] * e[2];
    return true;
  }
  static pgo::CostFunction* Create(double dx, double dy, double dtheta) {
    return new pgo::CostFunction{dx, dy, dtheta, pgo::CostFunction::SWITCHABLE};
  }
  double dx, dy, dtheta;
};

struct SwitchPriorResidue {
  explicit SwitchPriorResidue(double lambda_) : lambda(lambda_) {}
  template <typename T>
  bool operator()(const T* const S, T* e) const {
    e[0] = T(std::sqrt(lambda)) * (T(1.0) - S[0]);
    return true;
  }
  static pgo::CostFunction* Create(double lambda) { return new pgo::CostFunction{lambda, 0.0, 0.0, pgo::CostFunction::SWITCH_PRIOR}; }
  double lambda;
};

#endif
