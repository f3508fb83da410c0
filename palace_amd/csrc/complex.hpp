// Complex-valued layer: two real device vectors per complex vector, operators as (Ar, Ai) pairs.
// Follows palace/linalg/vector.hpp:23-147 (ComplexVector), linalg/operator.hpp:24-68 and
// linalg/operator.cpp:58-134 (ComplexOperator, ComplexWrapperOperator: up to four real applies per
// complex apply), linalg/vector.cpp:674-685 (Dot(x, y) = y^H x) and the complex instantiation of
// GmresSolver (linalg/iterative.cpp:543-705; the preconditioner is a real operator applied to the
// real and imaginary parts separately, linalg/gmg.cpp:147-168 `RealMult`).
#pragma once

#include <complex>

#include <memory>

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
// x[rows] = s ; x[rows] = y[rows] on both parts (vector.cpp:461-510)
void SetSubVector(const Context &c, ComplexVector &x, const int32_t *d_rows, int nrows, double s);
void SetSubVector(const Context &c, ComplexVector &x, const int32_t *d_rows, int nrows, const ComplexVector &y);
// x = conj(x) (vector.cpp Conj)
void Conj(const Context &c, ComplexVector &x);
// the remaining members of the reference's ComplexVector (linalg/vector.hpp:95-146, vector.cpp:172-460) as free functions:
// x^T y (no conjugate), x *= s, x = |x| (imaginary part zero), x = 1 ./ x, y = alpha x + beta y,
// z = alpha x + beta y + gamma z, x = [s_0 y_0; s_1 y_1; ...] (blocks laid end to end)
std::complex<double> TransposeDot(const Context &c, const ComplexVector &x, const ComplexVector &y);
void Scale(const Context &c, std::complex<double> s, ComplexVector &x);
void Abs(const Context &c, ComplexVector &x);
void Reciprocal(const Context &c, ComplexVector &x);
void AXPBY(const Context &c, std::complex<double> alpha, const ComplexVector &x, std::complex<double> beta, ComplexVector &y);
void AXPBYPCZ(const Context &c, std::complex<double> alpha, const ComplexVector &x, std::complex<double> beta,
              const ComplexVector &y, std::complex<double> gamma, ComplexVector &z);
void SetBlocks(const Context &c, ComplexVector &x, const std::vector<const ComplexVector *> &y,
               const std::vector<std::complex<double>> &s);
// complex instantiation of OrthogonalizeColumnMGS / CGS (orthog.hpp:41-89): H[j] = V[j]^H (W) w, w -= sum_j H[j] V[j];
// `weight` is a real operator applied to the real and the imaginary part (test/unit/test-orthog.cpp:49-67)
void OrthogonalizeColumn(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                         std::complex<double> *H, int m, const Operator *weight = nullptr);
// orthogonalise + norm + normalise with the coefficients on the device (orthog.hip; linalg.hpp: OrthonormalizeColumn)
double OrthonormalizeColumn(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                            std::complex<double> *H, int m);
void OrthogonalizeColumnDevice(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                               const ComplexVector *x, std::complex<double> *H, int m, bool normalize, double *hn);
}  // namespace linalg

// ComplexOperator (linalg/operator.hpp:24-68): abstract complex operator on ComplexVectors.  The variants a concrete
// operator does not provide abort like the reference's base class (operator.cpp:17-56).
class ComplexOperator {
protected:
  int height = 0, width = 0;

public:
  ComplexOperator(int s = 0) : height(s), width(s) {}
  ComplexOperator(int h, int w) : height(h), width(w) {}
  virtual ~ComplexOperator() = default;
  int Height() const { return height; }
  int Width() const { return width; }
  virtual bool IsReal() const { return !Imag(); }
  virtual bool IsImag() const { return !Real(); }
  virtual const Operator *Real() const { return nullptr; }
  virtual const Operator *Imag() const { return nullptr; }
  virtual void AssembleDiagonal(ComplexVector &diag) const;
  virtual void Mult(const ComplexVector &x, ComplexVector &y) const = 0;
  virtual void MultTranspose(const ComplexVector &x, ComplexVector &y) const;
  virtual void MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const;
  virtual void AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const;
  virtual void AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const;
  virtual void AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const;
};

// ComplexWrapperOperator (linalg/operator.hpp:70-111, operator.cpp:58-413): 2 x 2 real-equivalent form
//   [yr; yi] = [Ar -Ai; Ai Ar] [xr; xi]  over two real operators (either may be null), non-owning.
class ComplexWrapperOperator : public ComplexOperator {
  const Context *ctx_;
  const Operator *Ar_, *Ai_;
  bool fused_ = false;  // both parts are ceed::Operators with a one-pass complex form (pa_op_complex_fused)
  // ... or single-rank ParOperators around such a pair with the same essential dofs (real part: its policy; imaginary
  // part: DIAG_ZERO): the local operators and the policy of the fused masked apply
  const ceed::Operator *par_fused_r_ = nullptr, *par_fused_i_ = nullptr;
  int par_fused_policy_ = -1;
  mutable Vector t_, t2_;
  mutable ComplexVector tx_, ty_;
  // y (+)= s op(A) x for one real operator through whatever that operator offers (AddMult with a coefficient or
  // Mult + AXPY)
  void AddReal(const Operator *A, bool transpose, const Vector &x, Vector &y, double s) const;

public:
  ComplexWrapperOperator(const Context &ctx, const Operator *Ar, const Operator *Ai);
  const Operator *Real() const override { return Ar_; }
  const Operator *Imag() const override { return Ai_; }
  bool Fused() const { return fused_; }  // Mult is one pass over the element data for both parts
  void AssembleDiagonal(ComplexVector &diag) const override;
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
  void MultTranspose(const ComplexVector &x, ComplexVector &y) const override;
  void MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const override;
  void AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
  void AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
  void AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
};

// BaseProductOperator<ComplexOperator> (linalg/operator.hpp:270-352): y = A (B x), non-owning
class ComplexProductOperator : public ComplexOperator {
  const ComplexOperator &A_, &B_;
  mutable ComplexVector z_;

public:
  ComplexProductOperator(const ComplexOperator &A, const ComplexOperator &B)
      : ComplexOperator(A.Height(), B.Width()), A_(A), B_(B), z_(B.Height()) {}
  void Mult(const ComplexVector &x, ComplexVector &y) const override { B_.Mult(x, z_), A_.Mult(z_, y); }
  void MultTranspose(const ComplexVector &x, ComplexVector &y) const override {
    PA_REQUIRE(A_.Height() == A_.Width(), "the transposed product needs a square left factor (shared work vector)");
    A_.MultTranspose(x, z_), B_.MultTranspose(z_, y);
  }
  void MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const override {
    PA_REQUIRE(A_.Height() == A_.Width(), "the transposed product needs a square left factor (shared work vector)");
    A_.MultHermitianTranspose(x, z_), B_.MultHermitianTranspose(z_, y);
  }
  void AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override {
    B_.Mult(x, z_), A_.AddMult(z_, y, a);
  }
  void AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override {
    A_.MultTranspose(x, z_), B_.AddMultTranspose(z_, y, a);
  }
  void AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override {
    A_.MultHermitianTranspose(x, z_), B_.AddMultHermitianTranspose(z_, y, a);
  }
};

// BaseDiagonalOperator<ComplexOperator> (linalg/operator.hpp:354-423, operator.cpp:415-581): y = d .* x, non-owning
class ComplexDiagonalOperator : public ComplexOperator {
  const Context *ctx_;
  const ComplexVector &d_;
  void Apply(const ComplexVector &x, ComplexVector &y, std::complex<double> a, bool add, bool conj) const;

public:
  ComplexDiagonalOperator(const Context &ctx, const ComplexVector &d) : ComplexOperator(d.Size()), ctx_(&ctx), d_(d) {}
  void Mult(const ComplexVector &x, ComplexVector &y) const override { Apply(x, y, 1.0, false, false); }
  void MultTranspose(const ComplexVector &x, ComplexVector &y) const override { Apply(x, y, 1.0, false, false); }
  void MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const override { Apply(x, y, 1.0, false, true); }
  void AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override { Apply(x, y, a, true, false); }
  void AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override {
    Apply(x, y, a, true, false);
  }
  void AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override {
    Apply(x, y, a, true, true);
  }
};

// ComplexParOperator (linalg/rap.hpp:124-221, rap.cpp:393-749): y = P^T (Ar + i Ai) P x on true-dof vectors with the
// essential-dof handling done once on the complex vector (rows: DIAG_ONE copies x, DIAG_ZERO zeroes; the real part carries
// the policy, the imaginary part is DIAG_ZERO, rap.cpp:450-457).  Ar / Ai are the LOCAL operators (either may be null).
// Real() / Imag() expose the two real ParOperators the reference also keeps (RAPr / RAPi) for diagonal assembly and for
// preconditioner set-up.  With one rank Mult runs through them (essential masking fused into the element kernels, both
// parts of x in one pass over the operator data); the transposed forms run on the L-vectors exactly as written there.
class ComplexParOperator : public ComplexOperator {
  const Context *ctx_;
  const Operator *Ar_, *Ai_;
  const Halo *halo_;
  int n_true_, n_local_;
  std::unique_ptr<ComplexWrapperOperator> A_;  // local (L-vector) operator
  std::unique_ptr<ParOperator> RAPr_, RAPi_;
  std::unique_ptr<ComplexWrapperOperator> RAP_;  // wrapper over RAPr / RAPi (single-rank fast path)
  const ceed::Operator *fused_r_ = nullptr, *fused_i_ = nullptr;  // single rank: one pass for both parts (pa_op_mult_complex)
  void UpdateFused();
  int32_t *d_ess_ = nullptr;
  int n_ess_ = 0;
  ParOperator::DiagonalPolicy policy_ = ParOperator::DiagonalPolicy::DIAG_ONE;
  mutable ComplexVector lx_, ly_, tt_;
  void Prolongate(const ComplexVector &x, ComplexVector &lx) const;  // tx = x, tx[ess] = 0, lx = P tx
  void RestrictFix(const ComplexVector &x, ComplexVector &ly, ComplexVector &y) const;  // y = P^T ly, y[ess] = x | 0

public:
  ComplexParOperator(const Context &ctx, const Operator *Ar, const Operator *Ai, int n_true, const Halo *halo = nullptr);
  ~ComplexParOperator() override;
  // rap.cpp:436-462
  void SetEssentialTrueDofs(const int32_t *ess_host, int n_ess, ParOperator::DiagonalPolicy policy);
  const int32_t *GetEssentialTrueDofs() const { return d_ess_; }
  int NumEssentialTrueDofs() const { return n_ess_; }
  ParOperator::DiagonalPolicy GetDiagonalPolicy() const;
  const ComplexOperator &LocalOperator() const { return *A_; }
  const Operator *Real() const override { return RAPr_.get(); }
  const Operator *Imag() const override { return RAPi_.get(); }
  void AssembleDiagonal(ComplexVector &diag) const override;
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
  void MultTranspose(const ComplexVector &x, ComplexVector &y) const override;
  void MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const override;
  void AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
  void AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
  void AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a = 1.0) const override;
};

// Solver<ComplexOperator> (linalg/solver.hpp:21-65)
class ComplexSolver {
protected:
  int height = 0, width = 0;
  bool initial_guess = false;

public:
  virtual ~ComplexSolver() = default;
  int Height() const { return height; }
  virtual void SetOperator(const ComplexOperator &op) = 0;
  void SetInitialGuess(bool guess = true) { initial_guess = guess; }
  virtual void Mult(const ComplexVector &x, ComplexVector &y) const = 0;
  virtual void Mult2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const;
  virtual void MultTranspose2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const { Mult2(x, y, r); }
};

namespace linalg {
// power iteration for ||A||_2 (linalg/operator.cpp:583-631): herm = true iterates with A alone
double SpectralNorm(const Context &c, const ComplexOperator &A, bool herm, double tol = 1e-4, int max_it = 1000, uint64_t seed = 0);
}

// JacobiSmoother<ComplexOperator> (linalg/jacobi.cpp): y = D^-1 x with the complex diagonal
class ComplexJacobiSmoother : public ComplexSolver {
  const Context *ctx_;
  ComplexVector dinv_;

public:
  explicit ComplexJacobiSmoother(const Context &ctx) : ctx_(&ctx) {}
  void SetOperator(const ComplexOperator &op) override;
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
};

// ChebyshevSmoother<ComplexOperator> / ChebyshevSmoother1stKind<ComplexOperator> (linalg/chebyshev.cpp:160-293): the same
// polynomials in D^-1 A with the complex inverse diagonal; lambda_max by power iteration on D^-1 A (Hermitian iteration when
// the operator is real, chebyshev.cpp:22-28)
class ComplexChebyshevSmoother : public ComplexSolver {
  const Context *ctx_;
  int pc_it_, order_;
  double sf_max_, sf_min_, lambda_max_ = 0.0, theta_ = 0.0, delta_ = 0.0;
  bool fourth_kind_;
  const ComplexOperator *A_ = nullptr;
  ComplexVector dinv_;
  mutable ComplexVector d_, r_;

public:
  ComplexChebyshevSmoother(const Context &ctx, int smooth_it, int poly_order, double sf_max = 1.0, bool fourth_kind = true,
                           double sf_min = 0.0)
      : ctx_(&ctx), pc_it_(smooth_it), order_(poly_order), sf_max_(sf_max), sf_min_(sf_min), fourth_kind_(fourth_kind) {
    PA_REQUIRE(poly_order > 0, "Polynomial order for Chebyshev smoothing must be positive!");
  }
  void SetOperator(const ComplexOperator &op) override;
  double LambdaMax() const { return lambda_max_; }
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
  void Mult2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const override;
};

// DistRelaxationSmoother<ComplexOperator> (linalg/distrelaxation.cpp:14-151): Chebyshev on the complex Nedelec operator, then
// Chebyshev on the complex auxiliary (H1) operator through the real discrete gradient G applied to both parts
class ComplexDistRelaxationSmoother : public ComplexSolver {
  const Context *ctx_;
  int pc_it_;
  const Operator *G_;
  const ComplexOperator *A_ = nullptr;
  const ComplexParOperator *A_G_ = nullptr;
  std::unique_ptr<ComplexChebyshevSmoother> B_, B_G_;
  mutable ComplexVector x_G_, y_G_, r_G_, t_;

public:
  ComplexDistRelaxationSmoother(const Context &ctx, const Operator &G, int smooth_it, int cheby_smooth_it, int cheby_order,
                                double cheby_sf_max = 1.0, double cheby_sf_min = 0.0, bool cheby_4th_kind = true);
  void SetOperator(const ComplexOperator &) override { throw pa::Error("use SetOperators(op, op_G)"); }
  void SetOperators(const ComplexOperator &op, const ComplexParOperator &op_G);
  const ComplexChebyshevSmoother &Primary() const { return *B_; }
  const ComplexChebyshevSmoother &Auxiliary() const { return *B_G_; }
  void Mult(const ComplexVector &x, ComplexVector &y) const override { Mult2(x, y, t_); }
  void Mult2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const override;
  void MultTranspose2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const override;
};

// MfemWrapperSolver<ComplexOperator> (linalg/solver.hpp:67-120, solver.cpp): a real-valued solver applied to the real and
// the imaginary part of a complex vector; SetOperator hands it the real part of the complex operator (the reference's
// pc_mat_real construction for its coarse solvers).  Non-owning.
class ComplexWrapperSolver : public ComplexSolver {
  Solver *pc_;

public:
  explicit ComplexWrapperSolver(Solver &pc) : pc_(&pc) {}
  void SetOperator(const ComplexOperator &op) override {
    PA_REQUIRE(op.Real(), "the wrapped real solver needs the real part of the operator");
    height = op.Height(), width = op.Width();
    pc_->SetOperator(*op.Real());
  }
  void Mult(const ComplexVector &x, ComplexVector &y) const override {
    pc_->SetInitialGuess(initial_guess);
    pc_->Mult(x.Real(), y.Real());
    pc_->Mult(x.Imag(), y.Imag());
  }
};

// GeometricMultigridSolver<ComplexOperator> (linalg/gmg.cpp:16-205): complex operators and smoothers (Chebyshev, or the
// auxiliary-space smoother when the discrete gradients G are given) on every level, the real prolongations applied to both
// parts, the coarse solver any ComplexSolver (typically a ComplexWrapperSolver around a real one).  Levels 0 (coarsest) .. L-1.
class ComplexGeometricMultigridSolver : public ComplexSolver {
  const Context *ctx_;
  int pc_it_;
  std::vector<const Operator *> P_;
  std::vector<const ComplexParOperator *> A_;
  std::vector<std::unique_ptr<ComplexSolver>> B_;
  mutable std::vector<ComplexVector> X_, Y_, R_;
  void VCycle(int l, bool initial_guess) const;

public:
  ComplexGeometricMultigridSolver(const Context &ctx, std::unique_ptr<ComplexSolver> &&coarse_solver,
                                  const std::vector<const Operator *> &P, int cycle_it, int smooth_it, int cheby_order,
                                  double cheby_sf_max = 1.0, double cheby_sf_min = 0.0, bool cheby_4th_kind = true,
                                  const std::vector<const Operator *> *G = nullptr);
  void SetOperators(const std::vector<const ComplexParOperator *> &ops,
                    const std::vector<const ComplexParOperator *> *aux_ops = nullptr);
  void SetOperator(const ComplexOperator &) override { throw pa::Error("use SetOperators for multigrid"); }
  void Mult(const ComplexVector &x, ComplexVector &y) const override;
  const ComplexSolver &Smoother(int l) const { return *B_[l]; }
};

// GmresSolver<ComplexOperator> / FgmresSolver<ComplexOperator> (linalg/iterative.cpp:543-871): the shared implementation
// (krylov_impl.hpp) on ComplexVectors; the preconditioner is a real Solver applied to the real and the imaginary part
// (linalg/gmg.cpp:147-168 `RealMult`).
// Krylov solvers on ComplexOperators with a real preconditioner applied to both parts (or a complex one): the common part of
// IterativeSolver<ComplexOperator> (linalg/iterative.hpp:25-115)
class ComplexIterativeSolver {
protected:
  const Context *ctx_;
  const ComplexOperator *A_ = nullptr;
  const Solver *B_ = nullptr;          // real preconditioner applied to both parts, or
  const ComplexSolver *Bc_ = nullptr;  // a complex one
  double rel_tol_ = 0.0, abs_tol_ = 0.0;
  int max_it_ = 100, print_ = 0;
  mutable bool converged_ = false;
  mutable double initial_res_ = 1.0, final_res_ = 0.0;
  mutable int final_it_ = 0;
  void ApplyB(const ComplexVector &x, ComplexVector &y) const;  // iterative.cpp:243-256 (+ the "Preconditioner" phase range)

public:
  explicit ComplexIterativeSolver(const Context &ctx, int print = 0) : ctx_(&ctx), print_(print) {}
  virtual ~ComplexIterativeSolver() = default;
  void SetOperator(const ComplexOperator &op) { A_ = &op; }
  void SetPreconditioner(const Solver &pc) { B_ = &pc, Bc_ = nullptr; }
  void SetPreconditioner(const ComplexSolver &pc) { Bc_ = &pc, B_ = nullptr; }
  void SetTol(double t) { rel_tol_ = t; }
  void SetAbsTol(double t) { abs_tol_ = t; }
  void SetMaxIter(int n) { max_it_ = n; }
  virtual void Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess = false) const = 0;
  bool GetConverged() const { return converged_; }
  double GetInitialRes() const { return initial_res_; }
  double GetFinalRes() const { return final_res_; }
  int GetNumIterations() const { return final_it_; }
};

// GmresSolver / FgmresSolver<ComplexOperator> (iterative.cpp:543-871) over krylov_impl.hpp
class ComplexGmresSolver : public ComplexIterativeSolver {
  int max_dim_ = -1;
  bool flexible_ = false;
  PreconditionerSide pc_side_ = PreconditionerSide::LEFT;
  Orthogonalization orthog_ = Orthogonalization::MGS;
  mutable std::vector<ComplexVector> V_, Z_;
  mutable ComplexVector r_;

public:
  explicit ComplexGmresSolver(const Context &ctx, int print = 0, bool flexible = false)
      : ComplexIterativeSolver(ctx, print), flexible_(flexible),
        pc_side_(flexible ? PreconditionerSide::RIGHT : PreconditionerSide::LEFT) {}
  void SetRestartDim(int m) { max_dim_ = m; }
  void SetOrthogonalization(Orthogonalization o) { orthog_ = o; }
  void SetPreconditionerSide(PreconditionerSide side) {
    PA_REQUIRE(!flexible_ || side == PreconditionerSide::RIGHT, "FGMRES solver only supports right preconditioning!");
    pc_side_ = side;
  }
  void Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess = false) const override;
};

// CgSolver<ComplexOperator> (iterative.cpp:360-486): PCG for Hermitian positive definite systems, the scalars of the recurrence
// complex (inner products y^H x), the residual measured in the preconditioner's inner product.  Host-side scalars: one
// synchronisation per inner product, as in the reference.
class ComplexCgSolver : public ComplexIterativeSolver {
  mutable ComplexVector r_, z_, p_;

public:
  explicit ComplexCgSolver(const Context &ctx, int print = 0) : ComplexIterativeSolver(ctx, print) {}
  void Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess = false) const override;
};

}  // namespace palace
