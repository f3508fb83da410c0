// Restarted GMRES / FGMRES for real and complex scalars in one implementation, following the reference's
// GmresSolver<OperType>::Mult and FgmresSolver<OperType>::Mult statement by statement
// (palace/linalg/iterative.cpp:543-705 and :733-871) so that iteration counts and residual histories agree:
//   * left or right preconditioning for GMRES (InitialResidual / ApplyBA, iterative.cpp:184-241; the right-
//     preconditioned update is x += B (sum_k s_k V_k), :664-673), FGMRES = right preconditioning with the
//     preconditioned basis Z stored (:795, :824-827);
//   * the residual estimate from the Givens recursion, the restart logic and the exit conditions of the inner loop
//     (`converged || j + 1 == max_dim || it + 1 == max_it`, :638-643);
//   * plane rotations as LAPACK's d/zlartg (iterative.cpp:72-181).
// The scalar type, the vectors and the operator applies come from an `Ops` policy (real: linalg.hip, complex:
// complex.hip); nothing here touches the device directly.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <limits>
#include <vector>

#include "linalg.hpp"

namespace palace {
namespace krylov {

// ---- plane rotations (iterative.cpp:45-181) ----------------------------------------------------------------------
inline double SafeMin() {
  constexpr int fradix = std::numeric_limits<double>::radix;
  constexpr int expm = std::numeric_limits<double>::min_exponent, expM = std::numeric_limits<double>::max_exponent;
  return std::max(std::pow((double)fradix, (double)(expm - 1)), std::pow((double)fradix, (double)(1 - expM)));
}
inline double SafeMax() {
  constexpr int fradix = std::numeric_limits<double>::radix;
  constexpr int expm = std::numeric_limits<double>::min_exponent, expM = std::numeric_limits<double>::max_exponent;
  return std::min(std::pow((double)fradix, (double)(1 - expm)), std::pow((double)fradix, (double)(expM - 1)));
}

inline void GeneratePlaneRotation(const double dx, const double dy, double &cs, double &sn) {
  const double safmin = SafeMin(), safmax = SafeMax();
  const double root_min = std::sqrt(safmin), root_max = std::sqrt(safmax / 2);
  if (dy == 0.0) {
    cs = 1.0, sn = 0.0;
    return;
  }
  if (dx == 0.0) {
    cs = 0.0, sn = std::copysign(1.0, dy);
    return;
  }
  const double dx1 = std::abs(dx), dy1 = std::abs(dy);
  if (dx1 > root_min && dx1 < root_max && dy1 > root_min && dy1 < root_max) {
    const double d = std::sqrt(dx * dx + dy * dy);
    cs = dx1 / d;
    sn = dy / std::copysign(d, dx);
  } else {
    const double u = std::min(safmax, std::max(safmin, std::max(dx1, dy1)));
    const double dxs = dx / u, dys = dy / u;
    const double d = std::sqrt(dxs * dxs + dys * dys);
    cs = std::abs(dxs) / d;
    sn = dys / std::copysign(d, dx);
  }
}

inline void GeneratePlaneRotation(const std::complex<double> dx, const std::complex<double> dy, double &cs,
                                  std::complex<double> &sn) {
  // [ cs  sn; -conj(sn)  cs ] [dx; dy] = [r; 0], cs real (zlartg)
  using T = double;
  const T safmin = SafeMin(), safmax = SafeMax();
  if (dy == 0.0) {
    cs = 1.0, sn = 0.0;
    return;
  }
  if (dx == 0.0) {
    cs = 0.0;
    if (dy.real() == 0.0) {
      sn = std::conj(dy) / std::abs(dy.imag());
    } else if (dy.imag() == 0.0) {
      sn = std::conj(dy) / std::abs(dy.real());
    } else {
      const T root_min = std::sqrt(safmin), root_max = std::sqrt(safmax / 2);
      const T dy1 = std::max(std::abs(dy.real()), std::abs(dy.imag()));
      if (dy1 > root_min && dy1 < root_max) {
        sn = std::conj(dy) / std::sqrt(dy.real() * dy.real() + dy.imag() * dy.imag());
      } else {
        const T u = std::min(safmax, std::max(safmin, dy1));
        const std::complex<T> dys = dy / u;
        sn = std::conj(dys) / std::sqrt(dys.real() * dys.real() + dys.imag() * dys.imag());
      }
    }
    return;
  }
  const T root_min = std::sqrt(safmin), root_max = std::sqrt(safmax / 4);
  const T dx1 = std::max(std::abs(dx.real()), std::abs(dx.imag()));
  const T dy1 = std::max(std::abs(dy.real()), std::abs(dy.imag()));
  if (dx1 > root_min && dx1 < root_max && dy1 > root_min && dy1 < root_max) {
    const T dx2 = dx.real() * dx.real() + dx.imag() * dx.imag();
    const T dy2 = dy.real() * dy.real() + dy.imag() * dy.imag();
    const T dz2 = dx2 + dy2;
    if (dx2 >= dz2 * safmin) {
      cs = std::sqrt(dx2 / dz2);
      if (dx2 > root_min && dz2 < root_max * 2)
        sn = std::conj(dy) * (dx / std::sqrt(dx2 * dz2));
      else
        sn = std::conj(dy) * ((dx / cs) / dz2);
    } else {
      const T d = std::sqrt(dx2 * dz2);
      cs = dx2 / d;
      sn = std::conj(dy) * (dx / d);
    }
  } else {
    const T u = std::min(safmax, std::max(safmin, std::max(dx1, dy1)));
    T w;
    const std::complex<T> dys = dy / u;
    std::complex<T> dxs;
    const T dy2 = dys.real() * dys.real() + dys.imag() * dys.imag();
    T dx2, dz2;
    if (dx1 / u < root_min) {
      const T v = std::min(safmax, std::max(safmin, dx1));
      w = v / u;
      dxs = dx / v;
      dx2 = dxs.real() * dxs.real() + dxs.imag() * dxs.imag();
      dz2 = dx2 * w * w + dy2;
    } else {
      w = 1.0;
      dxs = dx / u;
      dx2 = dxs.real() * dxs.real() + dxs.imag() * dxs.imag();
      dz2 = dx2 + dy2;
    }
    if (dx2 >= dz2 * safmin) {
      cs = std::sqrt(dx2 / dz2);
      if (dx2 > root_min && dz2 < root_max * 2)
        sn = std::conj(dys) * (dxs / std::sqrt(dx2 * dz2));
      else
        sn = std::conj(dys) * ((dxs / cs) / dz2);
    } else {
      const T d = std::sqrt(dx2 * dz2);
      cs = dx2 / d;
      sn = std::conj(dys) * (dxs / d);
    }
    cs *= w;
  }
}

inline void ApplyPlaneRotation(double &dx, double &dy, const double cs, const double sn) {
  const double t = cs * dx + sn * dy;
  dy = -sn * dx + cs * dy;
  dx = t;
}
inline void ApplyPlaneRotation(std::complex<double> &dx, std::complex<double> &dy, const double cs,
                               const std::complex<double> sn) {
  const std::complex<double> t = cs * dx + sn * dy;
  dy = -std::conj(sn) * dx + cs * dy;
  dx = t;
}

enum class PreconditionerSide { LEFT = 0, RIGHT = 1 };

struct Result {
  bool converged = false;
  double initial_res = 1.0, final_res = 0.0;
  int final_it = 0;
};

struct Params {
  double rel_tol = 0.0, abs_tol = 0.0;
  int max_it = 100, max_dim = -1, print = 0;
  bool flexible = false, initial_guess = false;
  PreconditionerSide pc_side = PreconditionerSide::LEFT;
  Orthogonalization orthog = Orthogonalization::MGS;
  const char *name = "GMRES";
};

// Ops: { using Vec; using Scalar; int Size(); void Ensure(Vec &); void A(const Vec &, Vec &); bool HasB();
//        void B(const Vec &, Vec &); void Copy(const Vec &, Vec &); void Zero(Vec &); void BMinus(const Vec &b, Vec &r)
//        [r = b - r]; void Axpy(Scalar, const Vec &, Vec &); void Scale(double, Vec &); double Norm(const Vec &);
//        void Orthogonalize(Orthogonalization, const std::vector<Vec> &, Vec &w, Scalar *H, int m); }
template <class Ops>
void GmresMult(Ops &ops, const Params &p, const typename Ops::Vec &b, typename Ops::Vec &x,
               std::vector<typename Ops::Vec> &V, std::vector<typename Ops::Vec> &Z, typename Ops::Vec &r, Result &out) {
  using Vec = typename Ops::Vec;
  using Scalar = typename Ops::Scalar;
  const bool flexible = p.flexible;
  const bool haveB = ops.HasB();
  PA_REQUIRE(!flexible || haveB, "Operator and preconditioner must be set for FgmresSolver::Mult!");
  const PreconditionerSide side = flexible ? PreconditionerSide::RIGHT : p.pc_side;
  const bool left = haveB && side == PreconditionerSide::LEFT, right = haveB && side == PreconditionerSide::RIGHT;
  const int max_it = p.max_it, max_dim = p.max_dim < 0 ? p.max_it : p.max_dim;
  PA_REQUIRE(max_dim > 0, "GMRES restart dimension must be positive");
  if ((int)V.size() < max_dim + 1) V.resize(max_dim + 1);
  if (flexible && (int)Z.size() < max_dim + 1) Z.resize(max_dim + 1);
  ops.Ensure(r);
  ops.Ensure(V[0]);
  if (flexible) ops.Ensure(Z[0]);
  std::vector<Scalar> H((size_t)(max_dim + 1) * max_dim, Scalar(0.0)), s(max_dim + 1), sn(max_dim + 1);
  std::vector<double> cs(max_dim + 1);

  // InitialResidual (iterative.cpp:184-215): res <- B (b - A x) (left) or b - A x; `scratch` holds A x - b on the way
  auto initial_residual = [&](Vec &res, Vec &scratch, bool guess) {
    if (left) {
      if (guess) {
        ops.A(x, scratch);
        ops.BMinus(b, scratch);
        ops.B(scratch, res);
      } else {
        ops.B(b, res);
        ops.Zero(x);
      }
    } else {
      if (guess) {
        ops.A(x, res);
        ops.BMinus(b, res);
      } else {
        ops.Copy(b, res);
        ops.Zero(x);
      }
    }
  };

  double beta = 0.0, true_beta, eps = 0.0;
  out.converged = false;
  int it = 0, restart = 0;
  for (; it < max_it; restart++) {
    Vec &res = flexible ? Z[0] : r;
    initial_residual(res, V[0], p.initial_guess || restart > 0);
    true_beta = ops.Norm(res);
    PA_REQUIRE(std::isfinite(true_beta), "GMRES residual norm is not valid");
    if (it == 0) {
      if (p.initial_guess) {
        double beta_rhs;
        if (left) {
          ops.B(b, V[0]);
          beta_rhs = ops.Norm(V[0]);
        } else {
          beta_rhs = ops.Norm(b);
        }
        out.initial_res = beta_rhs;
      } else {
        out.initial_res = true_beta;
      }
      eps = std::max(p.rel_tol * out.initial_res, p.abs_tol);
    }
    beta = true_beta;
    if (beta < eps) {
      out.converged = true;
      break;
    }
    ops.Zero(V[0]);
    ops.Axpy(Scalar(1.0 / beta), res, V[0]);
    std::fill(s.begin(), s.end(), Scalar(0.0));
    s[0] = beta;

    int j = 0;
    for (;; j++, it++) {
      if (p.print > 1) std::printf("  %3d (restart %d) KSP residual norm %.6e\n", it, restart, beta);
      ops.Ensure(V[j + 1]);
      Vec &w = V[j + 1];
      // ApplyBA (iterative.cpp:217-241)
      if (left) {
        ops.A(V[j], r);
        ops.B(r, w);
      } else if (right) {
        Vec &z = flexible ? Z[j] : r;
        if (flexible) ops.Ensure(z);
        ops.B(V[j], z);
        ops.A(z, w);
      } else {
        ops.A(V[j], w);
      }
      Scalar *Hj = H.data() + (size_t)j * (max_dim + 1);
      ops.Orthogonalize(p.orthog, V, w, Hj, j + 1);
      const double hn = ops.Norm(w);
      Hj[j + 1] = hn;
      ops.Scale(1.0 / hn, w);
      for (int k = 0; k < j; k++) ApplyPlaneRotation(Hj[k], Hj[k + 1], cs[k], sn[k]);
      GeneratePlaneRotation(Hj[j], Hj[j + 1], cs[j], sn[j]);
      ApplyPlaneRotation(Hj[j], Hj[j + 1], cs[j], sn[j]);
      ApplyPlaneRotation(s[j], s[j + 1], cs[j], sn[j]);
      beta = std::abs(s[j + 1]);
      PA_REQUIRE(std::isfinite(beta), "GMRES residual norm is not valid");
      out.converged = beta < eps;
      if (out.converged || j + 1 == max_dim || it + 1 == max_it) {
        it++;
        break;
      }
    }
    // reconstruct the solution (restart, convergence or maximum iterations)
    for (int i = j; i >= 0; i--) {
      const Scalar *Hi = H.data() + (size_t)i * (max_dim + 1);
      s[i] /= Hi[i];
      for (int k = i - 1; k >= 0; k--) s[k] -= Hi[k] * s[i];
    }
    if (flexible) {
      for (int k = 0; k <= j; k++) ops.Axpy(s[k], Z[k], x);
    } else if (!right) {
      for (int k = 0; k <= j; k++) ops.Axpy(s[k], V[k], x);
    } else {
      ops.Zero(r);
      for (int k = 0; k <= j; k++) ops.Axpy(s[k], V[k], r);
      ops.B(r, V[0]);
      ops.Axpy(Scalar(1.0), V[0], x);
    }
    if (out.converged) break;
  }
  if (p.print > 1) std::printf("  %3d (restart %d) KSP residual norm %.6e\n", it, restart, beta);
  if (p.print > 0)
    std::printf("  %s solver %s in %d iteration%s (res %.3e, initial %.3e)\n", p.name,
                out.converged ? "converged" : "did NOT converge", it, it == 1 ? "" : "s", beta, out.initial_res);
  out.final_res = beta, out.final_it = it;
}

}  // namespace krylov
}  // namespace palace
