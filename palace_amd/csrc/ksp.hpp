// KspSolver: the linear solver object Palace's drivers hold -- a Krylov solver and its preconditioner configured from
// the "Linear" section of the configuration (reference linalg/ksp.hpp:36-86, ksp.cpp:27-333; utils/configfile.hpp:994-1122
// for LinearSolverData, utils/labels.hpp for the enumerations).
//
// Coarse-level solvers: the reference's choices on the coarsest multigrid level are external packages (HYPRE AMS /
// BoomerAMG, sparse direct solvers), which are outside this library.  Until a native auxiliary-space coarse solver
// exists two device-resident stand-ins are offered besides JACOBI: CHEBYSHEV_JACOBI (a fixed
// Chebyshev-Jacobi polynomial) and JACOBI_PCG (a few Jacobi-preconditioned CG iterations to a loose tolerance).
// Asking for AMS / BOOMER_AMG / a direct solver fails loudly.
#pragma once

#include <memory>

#include "complex.hpp"
#include "fem.hpp"
#include "linalg.hpp"

namespace palace {

enum class LinearSolver { DEFAULT, AMS, BOOMER_AMG, MUMPS, SUPERLU, STRUMPACK, STRUMPACK_MP, CUDSS, JACOBI,
                          CHEBYSHEV_JACOBI, JACOBI_PCG };
enum class KrylovSolver { DEFAULT, CG, MINRES, GMRES, FGMRES, BICGSTAB };
enum class PreconditionerSideOption { DEFAULT, RIGHT, LEFT };  // labels.hpp PreconditionerSide (with DEFAULT)
enum class MultigridCoarsening { LINEAR, LOGARITHMIC };

namespace config {

struct LinearSolverData {
  LinearSolver type = LinearSolver::DEFAULT;
  KrylovSolver krylov_solver = KrylovSolver::DEFAULT;
  double tol = 1.0e-6;    // iterative solver relative tolerance
  int max_it = 100;       // maximum number of iterations
  int max_size = -1;      // maximum Krylov space dimension (GMRES / FGMRES restart)
  int initial_guess = -1; // reuse the previous solution as the initial guess
  int mg_max_levels = -1;
  MultigridCoarsening mg_coarsening = MultigridCoarsening::LOGARITHMIC;
  bool mg_use_mesh = true;
  int mg_cycle_it = -1;   // V-cycles per preconditioner application
  int mg_smooth_aux = -1; // auxiliary-space (Hiptmair) smoothers
  int mg_smooth_it = 1;   // pre- / post-smoothing iterations
  int mg_smooth_order = -1;  // Chebyshev smoother order (-1: max(2 p, 4), iodata.cpp:533-564)
  double mg_smooth_sf_max = 1.0, mg_smooth_sf_min = 0.0;
  bool mg_smooth_cheby_4th = true;
  PreconditionerSideOption pc_side = PreconditionerSideOption::DEFAULT;
  Orthogonalization gs_orthog = Orthogonalization::MGS;
  // AMS (configfile.hpp:1087-1098): cycles per coarse solve, singular operator (no mass term: magnetostatics)
  int ams_max_it = -1;
  int ams_singular_op = -1;
  // stand-in coarse solvers only (not in the reference): iterations / tolerance of JACOBI_PCG, order of CHEBYSHEV_JACOBI
  int coarse_max_it = 8;
  double coarse_tol = 1.0e-2;
  int coarse_order = 4;
  // fills the -1 / DEFAULT entries the way IoData::CheckConfiguration does for an order-p problem of the given kind
  // (iodata.cpp:454-564): SPD problems (electrostatic, magnetostatic, transient) use CG and plain smoothers, the
  // frequency-domain ones GMRES and auxiliary-space smoothers
  void SetDefaults(int order, bool spd_problem);
};

}  // namespace config

// The coarsest level of a multi-rank hierarchy solved redundantly by every rank with the native algebraic cycles (where the
// reference runs HYPRE's distributed AMS / BoomerAMG on it: linalg/ams.cpp:18-224, amg.cpp:12-49, wiring ksp.cpp:129-239).
// Built from what each rank holds: its assembled local matrix (L-vector numbering, ghost rows with their partial sums), the
// halo plan of the space, its essential true dofs and -- AMS -- the discrete gradient operator and the coordinates of its true
// vertices.  True dofs are numbered rank by rank (offset of the rank + local true index), the ghosts learn their owners'
// numbers through the halo plan itself, the triplets / gradient rows / coordinates of all ranks are gathered with the
// communicator's global sum (every rank contributes zeros outside its own segment), and every rank assembles the SAME global
// matrix and builds the SAME solver on it; an application gathers the distributed right-hand side (ReplicatedSolver).
// G == nullptr: the AMG V-cycle (H1 problems); else the AMS cycle.  `level0` must be a ParOperator around an assembled
// (CsrOperator) or partially assembled local operator.
class ReplicatedCoarseSolver : public Solver {
  struct Impl;
  std::unique_ptr<Impl> impl_;

public:
  ReplicatedCoarseSolver(const Context &ctx, const Operator &level0, const Operator *G, int nv_true, const double *xyz_true, int dim,
                         int cycle_it, bool singular);
  ~ReplicatedCoarseSolver() override;
  void SetOperator(const Operator &) override {}
  void Mult(const Vector &x, Vector &y) const override;
  int GlobalSize() const;
  // Round 5: the SOLVE is distributed -- every rank keeps and applies its own rows of every level of the algebraic hierarchies
  // (amg_dist.hpp: DistAmgSolver / DistAmsSolver; the set-up still gathers the global matrix and builds the same hierarchy on
  // every rank); PALACE_AMD_COARSE_SOLVE=replicated at construction keeps the whole cycle on every rank (the gather + the
  // one-rank solvers, rounds 3-4)
  bool Distributed() const;
  const Solver *DistributedSolver() const;
};

// p-coarsening sequence of the multigrid hierarchy (fem/multigrid.hpp:44-69): orders from coarsest to finest
std::vector<int> GetPolynomialOrders(int order, MultigridCoarsening coarsening, int mg_max_levels = -1);

class KspSolver {
protected:
  std::unique_ptr<IterativeSolver> ksp;
  std::unique_ptr<Solver> pc;
  mutable int ksp_mult = 0, ksp_mult_it = 0;

public:
  KspSolver(const config::LinearSolverData &linear, int verbose, const FiniteElementSpaceHierarchy &fespaces,
            const FiniteElementSpaceHierarchy *aux_fespaces = nullptr);
  KspSolver(std::unique_ptr<IterativeSolver> &&ksp, std::unique_ptr<Solver> &&pc);
  int NumTotalMult() const { return ksp_mult; }
  int NumTotalMultIterations() const { return ksp_mult_it; }
  void SetRelTol(double tol) { ksp->SetTol(tol); }
  void SetAbsTol(double tol) { ksp->SetAbsTol(tol); }
  const IterativeSolver &GetKrylovSolver() const { return *ksp; }
  // op: the system operator; pc_op: the operator the preconditioner is built from (a MultigridOperator when the
  // preconditioner is geometric multigrid)
  void SetOperators(const Operator &op, const Operator &pc_op);
  void Mult(const Vector &x, Vector &y) const;
};

// BaseKspSolver<ComplexOperator>: CG (Hermitian positive definite systems) / GMRES / FGMRES on a ComplexOperator with a real-valued preconditioner applied to the
// real and the imaginary part (the reference's "PCMatReal" construction: the multigrid hierarchy is built from real
// operators, linalg/gmg.cpp:147-168 applies it part by part).  pc_op is that real operator (a MultigridOperator for
// geometric multigrid).
class ComplexKspSolver {
  std::unique_ptr<ComplexIterativeSolver> ksp;
  std::unique_ptr<Solver> pc;
  bool initial_guess = false;
  mutable int ksp_mult = 0, ksp_mult_it = 0;

public:
  ComplexKspSolver(const config::LinearSolverData &linear, int verbose, const FiniteElementSpaceHierarchy &fespaces,
                   const FiniteElementSpaceHierarchy *aux_fespaces = nullptr);
  int NumTotalMult() const { return ksp_mult; }
  int NumTotalMultIterations() const { return ksp_mult_it; }
  void SetRelTol(double tol) { ksp->SetTol(tol); }
  void SetAbsTol(double tol) { ksp->SetAbsTol(tol); }
  const ComplexIterativeSolver &GetKrylovSolver() const { return *ksp; }
  void SetOperators(const ComplexOperator &op, const Operator &pc_op);
  void Mult(const ComplexVector &x, ComplexVector &y) const;
};

}  // namespace palace
