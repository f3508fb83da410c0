// Complex-valued layer: two real device vectors per complex vector, operators as (Ar, Ai) pairs.
// Follows palace/linalg/vector.hpp:23-147 (ComplexVector), linalg/operator.hpp:24-68 and
// linalg/operator.cpp:58-134 (ComplexOperator, ComplexWrapperOperator: up to four real applies per
// complex apply), linalg/vector.cpp:674-685 (Dot(x, y) = y^H x) and the complex instantiation of
// GmresSolver (linalg/iterative.cpp:543-705; the preconditioner is a real operator applied to the
// real and imaginary parts separately, linalg/gmg.cpp:147-168 `RealMult`).
#pragma once

#include <complex>

#include "linalg.hpp"

namespace palace {

class ComplexVector {
  Vector xr_, xi_;

public:
  ComplexVector() = default;
  explicit ComplexVector(int n) : xr_(n), xi_(n) {}
  ComplexVector(double *re, double *im, int n) : xr_(re, n), xi_(im, n) {}
  void SetSize(int n) { xr_.SetSize(n), xi_.SetSize(n); }
  int Size() const { return xr_.Size(); }
  Vector &Real() { return xr_; }
  Vector &Imag() { return xi_; }
  const Vector &Real() const { return xr_; }
  const Vector &Imag() const { return xi_; }
};

namespace linalg {
std::complex<double> Dot(const Context &c, const ComplexVector &x, const ComplexVector &y);  // y^H x
double Norml2(const Context &c, const ComplexVector &x);
void AXPY(const Context &c, std::complex<double> alpha, const ComplexVector &x, ComplexVector &y);
void Scale(const Context &c, double s, ComplexVector &x);
void Copy(const Context &c, const ComplexVector &x, ComplexVector &y);
void Fill(const Context &c, ComplexVector &x, double s);
// complex instantiation of OrthogonalizeColumnMGS / CGS (orthog.hpp:41-89): H[j] = V[j]^H (W) w, w -= sum_j H[j] V[j];
// `weight` is a real operator applied to the real and the imaginary part (test/unit/test-orthog.cpp:49-67)
void OrthogonalizeColumn(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                         std::complex<double> *H, int m, const Operator *weight = nullptr);
}  // namespace linalg

class ComplexOperator {
protected:
  int height = 0, width = 0;

public:
  virtual ~ComplexOperator() = default;
  int Height() const { return height; }
  int Width() const { return width; }
  virtual void Mult(const ComplexVector &x, ComplexVector &y) const = 0;
};

class ComplexWrapperOperator : public ComplexOperator {
  const Context *ctx_;
  const Operator *Ar_, *Ai_;
  mutable Vector t_, t2_;

public:
  ComplexWrapperOperator(const Context &ctx, const Operator *Ar, const Operator *Ai);
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
};

// Restarted GMRES, left preconditioning, modified Gram-Schmidt, complex Givens rotations.
class ComplexGmresSolver {
  const Context *ctx_;
  const ComplexOperator *A_ = nullptr;
  const Solver *B_ = nullptr;  // real preconditioner, applied to both parts
  double rel_tol_ = 0.0, abs_tol_ = 0.0;
  int max_it_ = 100, max_dim_ = -1, print_ = 0;
  mutable bool converged_ = false;
  mutable double initial_res_ = 1.0, final_res_ = 0.0;
  mutable int final_it_ = 0;
  mutable std::vector<ComplexVector> V_;
  mutable ComplexVector r_;
  void ApplyB(const ComplexVector &x, ComplexVector &y) const;

public:
  explicit ComplexGmresSolver(const Context &ctx, int print = 0) : ctx_(&ctx), print_(print) {}
  void SetOperator(const ComplexOperator &op) { A_ = &op; }
  void SetPreconditioner(const Solver &pc) { B_ = &pc; }
  void SetTol(double t) { rel_tol_ = t; }
  void SetAbsTol(double t) { abs_tol_ = t; }
  void SetMaxIter(int n) { max_it_ = n; }
  void SetRestartDim(int m) { max_dim_ = m; }
  void Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess = false) const;
  bool GetConverged() const { return converged_; }
  double GetInitialRes() const { return initial_res_; }
  double GetFinalRes() const { return final_res_; }
  int GetNumIterations() const { return final_it_; }
};

}  // namespace palace
