// Native coarse-level solvers on the device: algebraic multigrid for H1-type matrices and the auxiliary-space (Hiptmair-Xu)
// preconditioner for H(curl) matrices built on it.
//
// They stand where the reference calls HYPRE on its coarsest multigrid level: BoomerAmgSolver (linalg/amg.cpp:12-49) and
// HypreAmsSolver (linalg/ams.cpp:18-224; wiring linalg/ksp.cpp:129-239).  HYPRE is third-party code outside /root/reference;
// what is restated here is the published algorithm it implements for these two calls and the reference's choice of options:
//   AMG   one V-cycle per application, polynomial / l1-Jacobi type smoothing (amg.cpp: relax type 18 on GPUs), a direct solve
//         on the last level -- over a smoothed-aggregation hierarchy (amg.hpp) instead of classical coarsening;
//   AMS   the multiplicative cycle 0 1 (3 + 4 + 5) 1 0 of ams.cpp's default cycle_type 14: smoothing on the edge matrix A,
//         a correction from the space of gradients (G^T A G, AMG), additive corrections from the three scalar nodal spaces
//         (Pi_c^T A Pi_c, AMG; Pi_c = the lowest-order Nedelec interpolation of a nodal field times e_c, built from G and the
//         vertex coordinates exactly as HYPRE_AMSSetCoordinateVectors does: Pi_c = |G| diag(G x_c) / 2), again gradients,
//         again smoothing.  `singular` (ams.cpp:28-30, :149-152, magnetostatics without a mass term) skips the gradient corrections.
// Set-up (aggregation, Galerkin products: amg.hpp) runs on the host once per matrix; every application runs on the device
// as sparse matrix-vector products and fused vector kernels on the context's stream, without host synchronisation, so it can
// sit inside a recorded V-cycle (StreamGraph).  One rank: the matrices are the rank's own (local) ones.
#pragma once

#include <memory>
#include <vector>

#include "amg.hpp"
#include "linalg.hpp"

namespace palace {

// device copy of a host CSR matrix and its operator
class DeviceCsr {
  pa_csr m_;
  std::unique_ptr<CsrOperator> op_;

public:
  DeviceCsr(const Context &ctx, const amg::HostCsr &h, bool symmetric);
  ~DeviceCsr();
  DeviceCsr(const DeviceCsr &) = delete;
  DeviceCsr &operator=(const DeviceCsr &) = delete;
  const CsrOperator &Op() const { return *op_; }
  int Rows() const { return m_.nrows; }
  long long Nnz() const { return m_.nnz; }
};

// host copy of a device CSR (pa_op_full_assemble's result) with ParOperator's essential rows / columns eliminated
// (diagonal 1: linalg/rap.cpp:131-149)
amg::HostCsr DownloadCsr(const pa_csr &m, const int32_t *ess_host = nullptr, int n_ess = 0);

struct AmgOptions {
  int max_levels = 12;
  int coarse_size = 400;    // rows of the level that is solved directly (dense pseudo-inverse)
  double theta = 0.08;      // strength threshold of the aggregation
  int smooth_order = 2;     // order of the 4th-kind Chebyshev smoother on D_l1^-1 A (1: one l1-Jacobi sweep)
};

class AmgSolver : public Solver {
  struct Level {
    std::unique_ptr<DeviceCsr> A, P, R;
    Vector dinv;  // 1 / sum_j |a_ij|: lambda_max(D_l1^-1 A) <= 1, no eigenvalue estimate needed
    mutable Vector x, b, r, d, t;
  };
  const Context *ctx_;
  AmgOptions opt_;
  bool fused_ = false;  // smoother steps and residuals in the sparse products' epilogues (CsrOperator::PrepareChebyStep, round 6)
  std::vector<Level> lv_;
  std::unique_ptr<DeviceCsr> Cinv_;  // pseudo-inverse of the last level's matrix (dense, stored as CSR)
  void Smooth(const Level &L, const Vector &b, Vector &x, bool zero_guess) const;
  void Cycle(size_t l, const Vector &b, Vector &x) const;

public:
  AmgSolver(const Context &ctx, const amg::HostCsr &A, const AmgOptions &opt = AmgOptions());
  void SetOperator(const Operator &) override {}  // the matrix is given at construction
  void Mult(const Vector &b, Vector &x) const override;  // x = B b: one V-cycle from a zero guess
  int NumLevels() const { return (int)lv_.size(); }
  int LevelRows(int l) const { return lv_[l].A->Rows(); }
  long long LevelNnz(int l) const { return lv_[l].A->Nnz(); }
  // host copies of the hierarchy (parity tests restate the cycle on them)
  const amg::Hierarchy &HostHierarchy() const { return host_; }
  const std::vector<double> &HostCoarseInverse() const { return host_cinv_; }

private:
  amg::Hierarchy host_;
  std::vector<double> host_cinv_;
};

struct AmsOptions {
  int cycle_it = 1;      // AMS cycles per application (ams.cpp: ams_it)
  int smooth_order = 2;  // Chebyshev order of the smoother on A (ams.cpp: one sweep of an l1 smoother)
  bool singular = false; // no mass term: skip the gradient-space corrections (ams.cpp:28-30, :149-152)
  AmgOptions amg;
};

class AmsSolver : public Solver {
  const Context *ctx_;
  AmsOptions opt_;
  bool fused_ = false;  // as in AmgSolver
  std::unique_ptr<DeviceCsr> A_, G_, Gt_, Pi_, Pit_;
  std::unique_ptr<AmgSolver> BG_, BPi_;
  Vector dinv_;
  mutable Vector r_, d_, t_, bg_, xg_, bp_, xp_;
  void Smooth(const Vector &b, Vector &x, bool zero_guess) const;
  void Correct(const DeviceCsr &T, const DeviceCsr &Tt, const AmgSolver &B, const Vector &b, Vector &x, Vector &bc,
               Vector &xc) const;

public:
  // A: the assembled H(curl) matrix (essential rows / columns eliminated), G: the discrete gradient [edges x vertices],
  // coords: vertex coordinates [nv][dim] (lowest order: ams.cpp:64-100); ess: the essential edge dofs (their rows of G and
  // Pi are dropped, so no correction touches them)
  AmsSolver(const Context &ctx, const amg::HostCsr &A, const amg::HostCsr &G, const double *coords, int dim,
            const std::vector<char> &ess_flag, const AmsOptions &opt = AmsOptions());
  void SetOperator(const Operator &) override {}
  void Mult(const Vector &b, Vector &x) const override;
  // (null for a singular operator: no gradient-space correction, ams.cpp:28-30)
  const AmgSolver *GradientSpaceSolver() const { return BG_.get(); }
  const AmgSolver *NodalSpaceSolver() const { return BPi_.get(); }
};

}  // namespace palace
