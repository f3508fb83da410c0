// The flux error estimators of linalg/errorestimator.{hpp,cpp} in 3-D: smooth flux recovery by a mass-matrix projection
// (FluxProjector, :111-187), element-wise error between the discontinuous and the smooth flux (ComputeErrorEstimates,
// :189-268), and the two estimators built on them (GradFluxErrorEstimator :271-360, CurlFluxErrorEstimator :390-510,
// TimeDependentFluxErrorEstimator :512-541), plus the running indicator they feed (fem/errorindicator.{hpp,cpp}).
// Spaces are dense-table spaces on one dense Mesh (fem.hpp); the operators are pa_op_add_sub_dense[_mixed] and
// pa_error_op_* (pa_mixed.hip).  Real vectors; a complex field is estimated part by part into the same estimates, as
// ComputeErrorEstimates does for a ComplexVector (:255-261).
#pragma once

#include <array>
#include <memory>
#include <vector>

#include "fem.hpp"
#include "ksp.hpp"

namespace palace {

namespace linalg {
// f(M) for a symmetric 3x3 matrix (column-major) through its eigen-decomposition: MatrixSqrt / MatrixPow of
// linalg/densematrix.cpp:222-252 as the estimators use them on the material tensors
std::array<double, 9> MatrixSqrt(const double *M);
std::array<double, 9> MatrixPow(const double *M, double p);
}  // namespace linalg

// fem/errorindicator.{hpp,cpp}: running root-mean-square of the element indicators over the solves of a simulation
class ErrorIndicator {
  const Context *ctx_;
  Vector local_;
  int n_ = 0;

public:
  explicit ErrorIndicator(const Context &ctx) : ctx_(&ctx) {}
  void AddIndicator(const Vector &indicator);  // errorindicator.cpp:11-47
  const Vector &Local() const { return local_; }
  double Norml2() const;
  int NumSamples() const { return n_; }
};

// what MaterialOperator hands to the estimators: attribute -> material index and one symmetric dim x dim tensor per material
// (mat_op.GetAttributeToMaterial() with GetPermittivityReal(), GetInvPermeability() or, for the scalar curl of a plane
// problem, the 1 x 1 GetCurlCurlInvPermeability(); column-major).  dim = 3 unless stated.
struct MaterialTensors {
  std::vector<int> attr_mat;
  std::vector<double> mat;  // [num_mat][dim * dim]
  int dim = 3;
  // f acts on symmetric 3 x 3 matrices: smaller tensors are bordered with an identity block, which f maps to itself
  // up to f(1) and which is dropped again
  template <typename F>
  MaterialTensors Map(F &&f) const {
    MaterialTensors out{attr_mat, mat, dim};
    const size_t dd = (size_t)dim * dim;
    for (size_t k = 0; k + dd <= mat.size(); k += dd) {
      double M[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
      for (int j = 0; j < dim; j++)
        for (int i = 0; i < dim; i++) M[i + 3 * j] = mat[k + i + (size_t)dim * j];
      const auto m = f(M);
      for (int j = 0; j < dim; j++)
        for (int i = 0; i < dim; i++) out.mat[k + i + (size_t)dim * j] = m[i + 3 * j];
    }
    return out;
  }
  MaterialPropertyCoefficient Coefficient() const { return MaterialPropertyCoefficient(attr_mat, dim, mat); }
};

// errorestimator.hpp:34-58, .cpp:111-187: y = M^-1 Flux x with M the mass matrix of the smooth space and Flux the
// coefficient-weighted mixed mass from the space of x into the smooth space; PCG + Jacobi (use_mg = false)
class FluxProjector {
  const Context *ctx_;
  std::unique_ptr<ceed::Operator> flux_, mass_;
  std::unique_ptr<ParOperator> M_;
  std::unique_ptr<JacobiSmoother> pc_;
  std::unique_ptr<CgSolver> pcg_;
  const FiniteElementSpace *smooth_, *rhs_space_;
  mutable Vector rhs_, lx_, ly_;

public:
  FluxProjector(const MaterialPropertyCoefficient &coeff, const FiniteElementSpace &smooth_fespace,
                const FiniteElementSpace &rhs_fespace, double tol, int max_it, int print);
  void Mult(const Vector &x, Vector &y) const;
  int NumIterations() const { return pcg_->GetNumIterations(); }
};

// Common part of the two estimators: F in `fespace`, its smooth recovery G in `smooth_fespace`, estimates += error^2
class FluxErrorEstimatorBase {
protected:
  const Context *ctx_;
  const FiniteElementSpace &fespace_, &smooth_fespace_;
  FluxProjector projector_;
  pa_error_op *integ_op_ = nullptr;
  mutable Vector G_;

  FluxErrorEstimatorBase(const FiniteElementSpace &fespace, const FiniteElementSpace &smooth_fespace,
                         const MaterialPropertyCoefficient &flux_coeff, int error_qf, const MaterialTensors &first,
                         const MaterialTensors &second, double tol, int max_it, int print);

public:
  virtual ~FluxErrorEstimatorBase();
  FluxErrorEstimatorBase(const FluxErrorEstimatorBase &) = delete;
  // ComputeErrorEstimates (:189-268): squared element errors added to `estimates` [num_elem]
  void AddErrorEstimates(const Vector &F, Vector &estimates) const;
  // AddErrorIndicator (:352-360, :502-510): sqrt(estimates) scaled by the total field energy
  void AddErrorIndicator(const Vector &F, double Et, ErrorIndicator &indicator) const;
  int NumElements() const { return fespace_.GetMesh().GetNE(); }
  const FluxProjector &GetProjector() const { return projector_; }
  const Vector &GetSmoothFlux() const { return G_; }
};

// eta_e^2 = || eps^-1/2 D - eps^1/2 E ||^2_e with D the RT recovery of eps E (E in ND)
class GradFluxErrorEstimator : public FluxErrorEstimatorBase {
public:
  GradFluxErrorEstimator(const MaterialTensors &epsilon, const FiniteElementSpace &nd_fespace,
                         const FiniteElementSpace &rt_fespace, double tol, int max_it, int print);
};

// eta_e^2 = || mu^1/2 H - mu^-1/2 B ||^2_e with H the ND recovery of mu^-1 B (B in RT).  Plane problems: B = curl E is a scalar
// in a discontinuous space, H its H1 recovery, muinv the 1 x 1 curl-curl tensor (errorestimator.cpp:446-472, f_apply_l2h1_error)
class CurlFluxErrorEstimator : public FluxErrorEstimatorBase {
public:
  CurlFluxErrorEstimator(const MaterialTensors &muinv, const FiniteElementSpace &rt_fespace,
                         const FiniteElementSpace &nd_fespace, double tol, int max_it, int print);
};

// :512-541: both of the above added before the square root
class TimeDependentFluxErrorEstimator {
  const Context *ctx_;
  GradFluxErrorEstimator grad_;
  CurlFluxErrorEstimator curl_;

public:
  TimeDependentFluxErrorEstimator(const MaterialTensors &epsilon, const MaterialTensors &muinv,
                                  const FiniteElementSpace &nd_fespace, const FiniteElementSpace &rt_fespace, double tol,
                                  int max_it, int print);
  void AddErrorIndicator(const Vector &E, const Vector &B, double Et, ErrorIndicator &indicator) const;
};

}  // namespace palace
