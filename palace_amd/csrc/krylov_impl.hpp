// Restarted GMRES / FGMRES for real and complex scalars in one implementation, following the reference's
// GmresSolver<OperType>::Mult and FgmresSolver<OperType>::Mult statement by statement
// (palace/linalg/iterative.cpp:543-705 and :733-871) so that iteration counts and residual histories agree:
//   * left or right preconditioning for GMRES (InitialResidual / ApplyBA, iterative.cpp:184-241; the right-
//     preconditioned update is x += B (sum_k s_k V_k), :664-673), FGMRES = right preconditioning with the
//     preconditioned basis Z stored (:795, :824-827);
//   * the residual estimate from the Givens recursion, the restart logic and the exit conditions of the inner loop
//     (`converged || j + 1 == max_dim || it + 1 == max_it`, :638-643);
//   * plane rotations: LAPACK's d/zlartg (safe scaling; the reference restates the same routine, iterative.cpp:72-181).
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

// ---- plane rotations -------------------------------------------------------------------------------------------------
// LAPACK's safe-scaling Givens rotation, la_lartg (LAPACK >= 3.10, SRC/dlartg.f90 / zlartg.f90; E. Anderson, "Algorithm 978: Safe
// scaling in the level 1 BLAS", ACM TOMS 44 (2017)) -- what the reference's GeneratePlaneRotation restates as well
// (linalg/iterative.cpp:45-181), so the Hessenberg recursion and with it the residual history agree to the last bit.  Written
// here once for both scalar types: f and g are brought into the range where |.|^2 neither over- nor underflows by a power-of-two
// independent scaling (u for both, a second one v for a much smaller f, undone by w = v / u), after which ONE set of formulas
// applies; without scaling u = w = 1 and the formulas are LAPACK's unscaled branch literally.
struct LartgRange {
  double tiny, huge;  // LAPACK's safmin / safmax: the smallest number whose reciprocal does not overflow, and that reciprocal
  LartgRange() {
    using lim = std::numeric_limits<double>;
    tiny = std::max(std::ldexp(1.0, lim::min_exponent - 1), std::ldexp(1.0, 1 - lim::max_exponent));
    huge = std::min(std::ldexp(1.0, 1 - lim::min_exponent), std::ldexp(1.0, lim::max_exponent - 1));
  }
  double Clamp(double a) const { return std::min(huge, std::max(tiny, a)); }
};
inline double MaxPart(double z) { return std::abs(z); }
inline double MaxPart(const std::complex<double> &z) { return std::max(std::abs(z.real()), std::abs(z.imag())); }
inline double SumSquares(double z) { return z * z; }
inline double SumSquares(const std::complex<double> &z) { return z.real() * z.real() + z.imag() * z.imag(); }

// [ c  s; -conj(s)  c ] [f; g] = [r; 0] with c real
inline void GeneratePlaneRotation(const double f, const double g, double &c, double &s) {
  static const LartgRange R;
  if (g == 0.0) {
    c = 1.0, s = 0.0;
    return;
  }
  if (f == 0.0) {
    c = 0.0, s = std::copysign(1.0, g);
    return;
  }
  const double lo = std::sqrt(R.tiny), hi = std::sqrt(R.huge / 2), af = std::abs(f), ag = std::abs(g);
  const bool safe = af > lo && af < hi && ag > lo && ag < hi;
  const double u = safe ? 1.0 : R.Clamp(std::max(af, ag));
  const double fs = safe ? f : f / u, gs = safe ? g : g / u;
  const double h = std::sqrt(fs * fs + gs * gs);
  c = std::abs(fs) / h;
  s = gs / std::copysign(h, f);
}

inline void GeneratePlaneRotation(const std::complex<double> f, const std::complex<double> g, double &c, std::complex<double> &s) {
  static const LartgRange R;
  if (g == 0.0) {
    c = 1.0, s = 0.0;
    return;
  }
  const double lo = std::sqrt(R.tiny), ag = MaxPart(g);
  if (f == 0.0) {  // r = |g|: only the modulus of g is needed, computed without overflow
    c = 0.0;
    if (g.real() == 0.0 || g.imag() == 0.0) {
      s = std::conj(g) / ag;
    } else {
      const double hi = std::sqrt(R.huge / 2);
      const std::complex<double> gs = (ag > lo && ag < hi) ? g : g / R.Clamp(ag);
      s = std::conj(gs) / std::sqrt(SumSquares(gs));
    }
    return;
  }
  const double hi = std::sqrt(R.huge / 4), af = MaxPart(f);
  std::complex<double> fs = f, gs = g;
  double w = 1.0;
  bool rescaled_f = false;
  if (!(af > lo && af < hi && ag > lo && ag < hi)) {
    const double u = R.Clamp(std::max(af, ag));
    gs = g / u;
    if (af / u < lo) {  // f much smaller than g: its own scaling, undone in c at the end
      const double v = R.Clamp(af);
      w = v / u, fs = f / v, rescaled_f = true;
    } else {
      fs = f / u;
    }
  }
  const double f2 = SumSquares(fs), g2 = SumSquares(gs), h2 = rescaled_f ? f2 * w * w + g2 : f2 + g2;
  if (f2 >= h2 * R.tiny) {
    c = std::sqrt(f2 / h2);
    s = std::conj(gs) * ((f2 > lo && h2 < hi * 2) ? fs / std::sqrt(f2 * h2) : (fs / c) / h2);
  } else {  // c would underflow through f2 / h2
    const double d = std::sqrt(f2 * h2);
    c = f2 / d;
    s = std::conj(gs) * (fs / d);
  }
  c *= w;
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
//        double Orthonormalize(Orthogonalization, const std::vector<Vec> &, Vec &w, Scalar *H, int m) [returns ||w|| before
//        the normalisation]; }
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
      // H(0 .. j, j) from the orthogonalisation, H(j+1, j) = ||w||, w /= H(j+1, j) (iterative.cpp:629-633): one call, so that the
      // scalars can stay on the device between the kernels (orthog.hip)
      Hj[j + 1] = ops.Orthonormalize(p.orthog, V, w, Hj, j + 1);
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
