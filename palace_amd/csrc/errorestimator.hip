// Flux error estimators (see errorestimator.hpp).  Host orchestration + two small element-wise kernels.
#include "errorestimator.hpp"

#include <hip/hip_runtime.h>

#include <cmath>
#include <limits>

namespace palace {

namespace {

void check(int rc) {
  if (rc) throw pa::Error(pa_last_error());
}

// errorindicator.cpp:41-43
__global__ void k_indicator_update(double *__restrict__ local, const double *__restrict__ ind, const int dn, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) local[i] = sqrt((local[i] * local[i] * dn + ind[i] * ind[i]) / (dn + 1));
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (the reference goes through MFEM's dense eigensolver,
// densematrix.cpp:150-220; any orthogonal diagonalisation gives the same f(M))
template <typename F>
std::array<double, 9> matrix_function(const double *M, F &&f) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = 0.5 * (M[i + 3 * j] + M[j + 3 * i]);
  for (int sweep = 0; sweep < 64; sweep++) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) {  // A <- A J
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq, A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A <- J^T A
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk, A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq, V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  std::array<double, 9> out{};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double v = 0.0;
      for (int k = 0; k < 3; k++) v += V[i][k] * f(A[k][k]) * V[j][k];
      out[i + 3 * j] = v;
    }
  return out;
}

}  // namespace

namespace linalg {

std::array<double, 9> MatrixSqrt(const double *M) {
  return matrix_function(M, [](double s) { return std::sqrt(s); });
}
std::array<double, 9> MatrixPow(const double *M, double p) {
  return matrix_function(M, [p](double s) { return std::pow(s, p); });
}

}  // namespace linalg

// ---- ErrorIndicator --------------------------------------------------------------------------------------------------
void ErrorIndicator::AddIndicator(const Vector &indicator) {
  if (n_ == 0) {
    local_.SetSize(indicator.Size());
    linalg::Copy(*ctx_, indicator, local_);
    n_ = 1;
    return;
  }
  PA_REQUIRE(local_.Size() == indicator.Size(), "Unexpected size mismatch for ErrorIndicator::AddIndicator!");
  const int n = local_.Size();
  if (n)
    hipLaunchKernelGGL(k_indicator_update, dim3((n + 255) / 256), dim3(256), 0, ctx_->stream, local_.Data(), indicator.Data(),
                       n_, n);
  n_ += 1;
}

double ErrorIndicator::Norml2() const { return n_ ? std::sqrt(linalg::Dot(*ctx_, local_, local_)) : 0.0; }

// ---- FluxProjector ---------------------------------------------------------------------------------------------------
FluxProjector::FluxProjector(const MaterialPropertyCoefficient &coeff, const FiniteElementSpace &smooth_fespace,
                             const FiniteElementSpace &rhs_fespace, double tol, int max_it, int print)
    : ctx_(&smooth_fespace.GetContext()), smooth_(&smooth_fespace), rhs_space_(&rhs_fespace) {
  PhaseRange range("Estimation / Construction");  // errorestimator.cpp:114
  // :117-120: a scalar smooth space (the H1 recovery of the scalar curl of a plane field) takes MassIntegrator
  const bool scalar_flux = smooth_fespace.GetFEType() == PA_FE_H1;
  {  // errorestimator.cpp:125-153 (use_mg = false): the mass matrix of the smooth space, no coefficient
    BilinearForm m(smooth_fespace);
    if (scalar_flux)
      m.AddDomainIntegrator<MassIntegrator>((const MaterialPropertyCoefficient *)nullptr);
    else
      m.AddDomainIntegrator<VectorFEMassIntegrator>((const MaterialPropertyCoefficient *)nullptr);
    mass_ = m.PartialAssemble();
    M_ = std::make_unique<ParOperator>(*ctx_, *mass_, smooth_fespace.GetTrueVSize(), nullptr, 0,
                                       ParOperator::DiagonalPolicy::DIAG_ONE, smooth_fespace.GetHalo());
  }
  {  // :154-176: the flux operator is always partially assembled
    BilinearForm flux(rhs_fespace, smooth_fespace);
    if (scalar_flux)
      flux.AddDomainIntegrator<MassIntegrator>(coeff);
    else
      flux.AddDomainIntegrator<VectorFEMassIntegrator>(coeff);
    flux_ = flux.PartialAssemble();
  }
  // ConfigureLinearSolver (:66-107): the system matrix is real, SPD and diagonally dominant
  pc_ = std::make_unique<JacobiSmoother>(*ctx_);
  pcg_ = std::make_unique<CgSolver>(*ctx_, print);
  pcg_->SetInitialGuess(false);
  pcg_->SetTol(tol);
  pcg_->SetAbsTol(std::numeric_limits<double>::epsilon());
  pcg_->SetMaxIter(max_it);
  pcg_->SetOperator(*M_);
  pc_->SetOperator(*M_);
  pcg_->SetPreconditioner(*pc_);
  rhs_.SetSize(smooth_fespace.GetTrueVSize());
  if (rhs_fespace.GetHalo()) lx_.SetSize(rhs_fespace.GetVSize());
  if (smooth_fespace.GetHalo()) ly_.SetSize(smooth_fespace.GetVSize());
}

void FluxProjector::Mult(const Vector &x, Vector &y) const {
  PhaseRange range("Estimation / Solve");  // errorestimator.cpp:172
  PA_REQUIRE(x.Size() == rhs_space_->GetTrueVSize() && y.Size() == rhs_.Size(), "Invalid vector dimensions for FluxProjector::Mult!");
  // Flux as a ParOperator between two spaces (rap.cpp:207-220 without essential dofs): P_test^T A P_trial
  const Halo *hx = rhs_space_->GetHalo(), *hy = smooth_->GetHalo();
  const Vector *in = &x;
  if (hx) {
    Vector t(lx_.Data(), x.Size());
    linalg::Copy(*ctx_, x, t);
    hx->Prolongate(lx_.Data(), ctx_->stream);
    in = &lx_;
  }
  if (hy) {
    flux_->Mult(*in, ly_);
    hy->RestrictAdd(ly_.Data(), ctx_->stream);
    Vector t(ly_.Data(), rhs_.Size());
    linalg::Copy(*ctx_, t, rhs_);
  } else {
    flux_->Mult(*in, rhs_);
  }
  pcg_->Mult(rhs_, y);
}

// ---- estimators ------------------------------------------------------------------------------------------------------
FluxErrorEstimatorBase::FluxErrorEstimatorBase(const FiniteElementSpace &fespace, const FiniteElementSpace &smooth_fespace,
                                               const MaterialPropertyCoefficient &flux_coeff, int error_qf,
                                               const MaterialTensors &first, const MaterialTensors &second, double tol,
                                               int max_it, int print)
    : ctx_(&fespace.GetContext()), fespace_(fespace), smooth_fespace_(smooth_fespace),
      projector_(flux_coeff, smooth_fespace, fespace, tol, max_it, print), G_(smooth_fespace.GetTrueVSize()) {
  PA_REQUIRE(fespace.IsDense() && smooth_fespace.IsDense() && &fespace.GetMesh() == &smooth_fespace.GetMesh(),
             "the estimators take two dense-table spaces on one mesh");
  const auto c1 = first.Coefficient(), c2 = second.Coefficient();
  PA_REQUIRE(first.dim == second.dim, "the two coefficients of an error integrator have one dimension");
  const auto ctx = ceed::PopulateCoefficientContext(first.dim, &c1, second.dim, &c2);  // errorestimator.cpp:326-329, :459-460
  const auto r1 = fespace.GetCeedElemRestriction(), r2 = smooth_fespace.GetCeedElemRestriction();
  const auto b1 = fespace.GetCeedDenseBasis(), b2 = smooth_fespace.GetCeedDenseBasis();
  check(pa_error_op_create(fespace.GetMesh().GetCeedGeomFactorData(), &r1, &b1, &r2, &b2, error_qf, ctx.data(),
                           ctx.size() * sizeof(double), &integ_op_));
}

FluxErrorEstimatorBase::~FluxErrorEstimatorBase() { pa_error_op_destroy(integ_op_); }

void FluxErrorEstimatorBase::AddErrorEstimates(const Vector &F, Vector &estimates) const {
  PA_REQUIRE(F.Size() == fespace_.GetTrueVSize() && estimates.Size() == fespace_.GetMesh().GetNE(),
             "Invalid vector dimensions for the error estimate!");
  projector_.Mult(F, G_);
  // grid functions = L-vectors (GetProlongationMatrix()->Mult, :202-214): ghosts filled through the halos
  const Halo *hf = fespace_.GetHalo(), *hg = smooth_fespace_.GetHalo();
  Vector F_gf, G_gf;
  const double *pf = F.Data(), *pg = G_.Data();
  if (hf) {
    F_gf.SetSize(fespace_.GetVSize());
    Vector t(F_gf.Data(), F.Size());
    linalg::Copy(*ctx_, F, t);
    hf->Prolongate(F_gf.Data(), ctx_->stream);
    pf = F_gf.Data();
  }
  if (hg) {
    G_gf.SetSize(smooth_fespace_.GetVSize());
    Vector t(G_gf.Data(), G_.Size());
    linalg::Copy(*ctx_, G_, t);
    hg->Prolongate(G_gf.Data(), ctx_->stream);
    pg = G_gf.Data();
  }
  check(pa_error_op_apply_add(integ_op_, pf, pg, estimates.Data(), ctx_->stream));
  if (hf || hg) PA_HIP(hipStreamSynchronize(ctx_->stream));  // the temporaries go out of scope
}

void FluxErrorEstimatorBase::AddErrorIndicator(const Vector &F, double Et, ErrorIndicator &indicator) const {
  PhaseRange range("Estimation");  // errorestimator.cpp:192
  Vector estimates(fespace_.GetMesh().GetNE());
  linalg::Fill(*ctx_, estimates, 0.0);
  AddErrorEstimates(F, estimates);
  linalg::Sqrt(*ctx_, estimates, (Et > 0.0) ? 0.5 / Et : 1.0);  // Correct factor of 1/2 in energy
  indicator.AddIndicator(estimates);
}

GradFluxErrorEstimator::GradFluxErrorEstimator(const MaterialTensors &epsilon, const FiniteElementSpace &nd_fespace,
                                               const FiniteElementSpace &rt_fespace, double tol, int max_it, int print)
    : FluxErrorEstimatorBase(nd_fespace, rt_fespace, epsilon.Coefficient(),
                             nd_fespace.GetMesh().Dimension() == 2 ? PA_QF_HCURLHDIV_ERROR_22 : PA_QF_HCURLHDIV_ERROR_33,  // :343-356
                             epsilon.Map([](const double *m) { return linalg::MatrixSqrt(m); }),
                             epsilon.Map([](const double *m) { return linalg::MatrixPow(m, -0.5); }), tol, max_it, print) {}

CurlFluxErrorEstimator::CurlFluxErrorEstimator(const MaterialTensors &muinv, const FiniteElementSpace &rt_fespace,
                                               const FiniteElementSpace &nd_fespace, double tol, int max_it, int print)
    : FluxErrorEstimatorBase(rt_fespace, nd_fespace, muinv.Coefficient(),
                             rt_fespace.GetMesh().Dimension() == 2 ? PA_QF_L2H1_ERROR : PA_QF_HDIVHCURL_ERROR_33,  // :464-480
                             muinv.Map([](const double *m) { return linalg::MatrixSqrt(m); }),
                             muinv.Map([](const double *m) { return linalg::MatrixPow(m, -0.5); }), tol, max_it, print) {}

TimeDependentFluxErrorEstimator::TimeDependentFluxErrorEstimator(const MaterialTensors &epsilon, const MaterialTensors &muinv,
                                                                 const FiniteElementSpace &nd_fespace,
                                                                 const FiniteElementSpace &rt_fespace, double tol,
                                                                 int max_it, int print)
    : ctx_(&nd_fespace.GetContext()), grad_(epsilon, nd_fespace, rt_fespace, tol, max_it, print),
      curl_(muinv, rt_fespace, nd_fespace, tol, max_it, print) {}

void TimeDependentFluxErrorEstimator::AddErrorIndicator(const Vector &E, const Vector &B, double Et,
                                                        ErrorIndicator &indicator) const {
  Vector estimates(grad_.NumElements());
  linalg::Fill(*ctx_, estimates, 0.0);
  grad_.AddErrorEstimates(E, estimates);
  curl_.AddErrorEstimates(B, estimates);  // grad_estimates += curl_estimates (:536)
  linalg::Sqrt(*ctx_, estimates, (Et > 0.0) ? 0.5 / Et : 1.0);  // Correct factor of 1/2 in energy
  indicator.AddIndicator(estimates);
}

}  // namespace palace
