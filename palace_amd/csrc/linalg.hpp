// Host-side C++ mirror of the pieces of palace::linalg that sit on the hot path, written against
// device (HBM) vectors and HIP streams.  Names, argument meaning and conventions follow the
// reference so Palace's drivers can use these objects in place of its own:
//   Operator / Solver            palace/linalg/operator.hpp:21, palace/linalg/solver.hpp:21-65
//   ceed::Operator               palace/fem/libceed/operator.hpp:32-65
//   ParOperator                  palace/linalg/rap.hpp, rap.cpp:154-234
//   linalg:: vector kernels      palace/linalg/vector.cpp:276-592,665-698, vector.hpp:247-270
//   CgSolver                     palace/linalg/iterative.cpp:360-486
//   GmresSolver                  palace/linalg/iterative.cpp:543-705 (MGS, orthog.hpp:41-55)
//   ChebyshevSmoother (4th kind) palace/linalg/chebyshev.cpp:160-220, 1st kind :222-293
//   JacobiSmoother               palace/linalg/jacobi.cpp:74-104
//   GeometricMultigridSolver     palace/linalg/gmg.cpp:16-205
//   SpectralNorm                 palace/linalg/operator.cpp:583-631
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "pa_internal.hpp"

namespace palace {

class Comm;  // RCCL communicator (comm.hpp); nullptr = single process
class Halo;  // conforming prolongation of one space across ranks (comm.hpp)
struct HaloStep;

// Reduction scratch of one context (per-block partial sums on the device, a pinned slot for the results); created on
// first use and shared by the copies of a context -- never by two contexts, whose streams may reduce concurrently.
class Workspace {
  double *d_ = nullptr, *h_ = nullptr, *gs_d_ = nullptr, *gs_h_ = nullptr;
  size_t gs_dn_ = 0, gs_hn_ = 0;
  hipStream_t halo_stream_ = nullptr;
  hipEvent_t ev_ready_ = nullptr, ev_done_ = nullptr;

public:
  static constexpr size_t kDeviceDoubles = 32768, kPinnedDoubles = 64;  // fixed: recorded graphs keep these addresses
  Workspace() = default;
  Workspace(const Workspace &) = delete;
  Workspace &operator=(const Workspace &) = delete;
  ~Workspace();
  double *Device(size_t n);  // n <= kDeviceDoubles doubles of device memory
  double *Pinned(size_t n);  // n <= kPinnedDoubles doubles of page-locked host memory
  // scratch of the device-side Gram-Schmidt (orthog.hip): grows with the restart length, zero-filled when (re)allocated;
  // never part of a recorded sequence
  double *GsDevice(size_t n);
  double *GsPinned(size_t n);
  // second stream for halo exchanges that overlap with interior element work, and the two events of the fork / join
  // (ready: the vector to exchange is complete on the main stream; done: the ghosts have arrived on the halo stream)
  hipStream_t HaloStream();
  hipEvent_t ReadyEvent();
  hipEvent_t DoneEvent();
};

// Execution context shared by the objects of one solve: the stream everything is enqueued on and
// the communicator used by global reductions / halo exchanges.
struct Context {
  hipStream_t stream = nullptr;
  Comm *comm = nullptr;
  mutable std::shared_ptr<Workspace> ws;
  Workspace &Work() const {
    if (!ws) ws = std::make_shared<Workspace>();
    return *ws;
  }
};

// A launch sequence recorded once as a HIP graph and replayed with one hipGraphLaunch: for the fixed-shape inner
// loops (one PCG iteration, one multigrid V-cycle) whose ~100 small launches cost more host time than device time
// on the per-rank problem sizes of a strong-scaling run.  `body` must enqueue on c.stream only, must not wait for
// the device, and must use the same buffers every time it runs with the same key (the pointers the caller passes in);
// its first run with a key is direct (work vectors get their sizes there), the second one is captured.  A failed
// capture disables the graph for good and the body runs directly; so does a context on the null stream, which cannot
// be recorded.  PALACE_AMD_GRAPH=0 disables all graphs.
//
// Staleness: a recording bakes in kernel arguments (operator and smoother device pointers, Chebyshev lambda_max, essential
// lists, tolerances).  Every call that re-configures an operator or solver in place (SetOperator(s), SetPreconditioner,
// SetEssential, SetTol/SetMaxIter, coefficient changes of a SumOperator, ...) calls StreamGraph::Invalidate(), which bumps a
// process-wide configuration epoch; a recording made under an older epoch is dropped and made again on its next Run.
class StreamGraph {
  hipGraphExec_t exec_ = nullptr;
  std::vector<const void *> key_;
  unsigned long long epoch_ = 0;
  int seen_ = 0;
  bool disabled_ = false;
  bool Capture(const Context &c, const std::function<void()> &body);

public:
  StreamGraph();
  StreamGraph(const StreamGraph &) = delete;
  StreamGraph &operator=(const StreamGraph &) = delete;
  ~StreamGraph();
  void Reset();
  bool Captured() const { return exec_ != nullptr; }
  // true while this thread records a sequence; code that has to wait for the device calls RequireNotRecording first
  // (it throws, the recording is dropped cleanly and the sequence runs directly from then on)
  static bool Recording();
  static void RequireNotRecording(const char *what);
  static void Invalidate();  // some operator/solver was re-configured: every recording is stale
  static unsigned long long Epoch();
  void Run(const Context &c, const std::vector<const void *> &key, const std::function<void()> &body);
};

// Named host ranges for profilers, after the reference's BlockTimer phases (utils/timer.hpp:29-56, :59-84): "Linear Solve",
// "  Setup", "  Preconditioner", "  Coarse Solve", "Operator Construction", "Estimation" ... -- emitted as roctx ranges
// (rocprofv3 --marker-trace shows them over the kernels of the phase).  The roctx library is looked up at run time on
// first use; without it, or with PALACE_AMD_ROCTX=0, a range costs one branch.  Ranges inside a recorded sequence appear on
// the runs that execute the host code (the first two), not on graph replays.
class PhaseRange {
  bool on_;

public:
  explicit PhaseRange(const char *name);
  ~PhaseRange();
  PhaseRange(const PhaseRange &) = delete;
  PhaseRange &operator=(const PhaseRange &) = delete;
};

// Device vector: owning, or a view of caller memory (mfem::Vector with device memory in Palace).
class Vector {
  double *d_ = nullptr;
  int n_ = 0;
  bool own_ = false;

public:
  Vector() = default;
  explicit Vector(int n) { SetSize(n); }
  Vector(double *ext, int n) : d_(ext), n_(n), own_(false) {}
  Vector(const Vector &) = delete;
  Vector &operator=(const Vector &) = delete;
  Vector(Vector &&o) noexcept : d_(o.d_), n_(o.n_), own_(o.own_) { o.d_ = nullptr, o.n_ = 0, o.own_ = false; }
  Vector &operator=(Vector &&o) noexcept;
  ~Vector();
  void SetSize(int n);
  void MakeRef(double *ext, int n);
  int Size() const { return n_; }
  double *Data() { return d_; }
  const double *Data() const { return d_; }
};

namespace linalg {

// y = x ; x = s
void Copy(const Context &c, const Vector &x, Vector &y);
void Fill(const Context &c, Vector &x, double s);
// vector.cpp:702-785
void AXPY(const Context &c, double alpha, const Vector &x, Vector &y);
void AXPBY(const Context &c, double alpha, const Vector &x, double beta, Vector &y);
void AXPBYPCZ(const Context &c, double alpha, const Vector &x, double beta, const Vector &y, double gamma,
              Vector &z);
// x[rows] = s ; x[rows] = y[rows]   (vector.cpp:461-510)
void SetSubVector(const Context &c, Vector &x, const int32_t *d_rows, int nrows, double s);
void SetSubVector(const Context &c, Vector &x, const int32_t *d_rows, int nrows, const Vector &y);
// y = x .* y, x = 1 ./ x, x *= s
void Scale(const Context &c, const Vector &d, Vector &y);
void Scale(const Context &c, double s, Vector &x);
void Reciprocal(const Context &c, Vector &x);
// Global inner product of T-vectors: local wavefront/LDS tree reduction + allreduce
// (vector.hpp:247-260).
double Dot(const Context &c, const Vector &x, const Vector &y);
double Norml2(const Context &c, const Vector &x);
// global sum of the entries (vector.hpp Sum -> LocalSum, vector.cpp:687-699) and x = sqrt(s x) (vector.cpp:774-781)
double Sum(const Context &c, const Vector &x);
void Sqrt(const Context &c, Vector &x, double s = 1.0);
// x /= ||x||, returns the norm (vector.hpp:264-270)
double Normalize(const Context &c, Vector &x);
// Deterministic uniform [-1, 1) fill from a counter-based generator (stands in for
// SetRandom, vector.cpp:595-605; the reference's stream comes from MFEM and is not reproducible)
void SetRandom(const Context &c, Vector &x, uint64_t seed);
// chebyshev.cpp:69-156
void ChebyOrder0(const Context &c, double sr, const Vector &dinv, const Vector &r, Vector &d);
void ChebyOrderK(const Context &c, double sd, double sr, const Vector &dinv, const Vector &r, Vector &d);
// fused forms of consecutive reference kernels (same arithmetic, one pass):
//   ChebyStep: y += d; r -= t; d = sd d + sr dinv .* r      (chebyshev.cpp:208-216, t = A d)
//   CgUpdate : x += a p; r -= a z                            (iterative.cpp:448-449)
//   ChebyStep3: out (+)= e_k + sd (e_k - e_prev) + sr dinv .* (r0 - t), t = A e_k  (the same step for the accumulated correction;
//   e_prev == nullptr: zero)
void ChebyStep3(const Context &c, double sd, double sr, const Vector &dinv, const Vector &t, const Vector &r0, const Vector &ek,
                const Vector *e_prev, Vector &out, bool add);
void ChebyStep(const Context &c, double sd, double sr, const Vector &dinv, const Vector &t, Vector &r, Vector &d,
               Vector &y);
void CgUpdate(const Context &c, double a, const Vector &p, const Vector &z, Vector &x, Vector &r);

}  // namespace linalg

// mfem::Operator as Palace uses it (linalg/operator.hpp:21).
class Operator {
protected:
  int height = 0, width = 0;

public:
  Operator(int h = 0, int w = 0) : height(h), width(w) {}
  virtual ~Operator() = default;
  int Height() const { return height; }
  int Width() const { return width; }
  virtual void Mult(const Vector &x, Vector &y) const = 0;
  // y = A^T x: no default (an operator that is not known to be symmetric must say what its transpose is)
  virtual void MultTranspose(const Vector &x, Vector &y) const;
  // y += a A x, y += a A^T x
  virtual void AddMult(const Vector &x, Vector &y, double a = 1.0) const;
  virtual void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const;
  virtual void AssembleDiagonal(Vector &diag) const;
  // true when A^T = A is known (symmetric coefficients): lets wrappers forward the transpose to the fused forward paths,
  // like the reference's SymmetricOperator (fem/libceed/operator.hpp:69-79)
  virtual bool IsSymmetric() const { return false; }
  // One step of the accumulated Chebyshev recurrence (chebyshev.cpp:204-218; linalg::ChebyStep3) with t = A x consumed where the
  // operator produces it:  out (+)= x + sd (x - e_prev) + sr dinv .* (r0 - A x).  PrepareChebyStep (set-up time, never inside a
  // recorded sequence): true when the operator has such a form; otherwise the smoother applies A and runs the vector kernel.
  struct ChebyStepArgs {
    double sd, sr;
    const Vector *dinv, *r0, *e_prev;  // e_prev == nullptr: zero
    Vector *out;
    bool add;
  };
  virtual bool PrepareChebyStep() const { return false; }
  virtual void MultChebyStep(const Vector &x, const ChebyStepArgs &a) const;
  // res = b - A y and / or d0 = c0 dinv .* (b - A y) in the same place (operators with PrepareChebyStep() only)
  virtual void MultResidual(const Vector &y, const Vector &b, Vector *res, const Vector *dinv = nullptr, double c0 = 0.0,
                            Vector *d0 = nullptr) const;
};

namespace ceed {

// palace::ceed::Operator over the C ABI (non-owning or owning handle).
class Operator : public palace::Operator {
  pa_op *op_;
  bool own_;
  const Context *ctx_;

  Vector dof_multiplicity_;  // SetDofMultiplicity (operator.hpp:35, :54): empty = none
  mutable Vector temp_;

public:
  Operator(const Context &ctx, pa_op *op, bool own);
  // Operator::Operator(h, w) (fem/libceed/operator.cpp:17-42): an empty composite, filled by AddSubOperator, closed by Finalize
  Operator(const Context &ctx, int h, int w);
  ~Operator() override;
  pa_op *Handle() const { return op_; }
  // AddSubOperator (operator.cpp:60-87).  The reference hands over a CeedOperator built by AssembleCeedOperator
  // (fem/libceed/integrator.cpp:423-513) from (geometry data, restriction, basis, QFunction, context); here the same five
  // things are the sub-operator (tensor or dense basis tables).  The transpose sub-operator of the reference's second
  // argument is implied: every sub-operator carries its transposed form (pa_op_mult_transpose).
  void AddSubOperator(pa_geom *geom, const pa_restriction_desc &restr, const pa_basis_desc &basis, int qfunction, const void *qf_ctx,
                      size_t ctx_size, uint32_t trial_ops, uint32_t test_ops);
  void AddSubOperator(pa_geom *geom, const pa_restriction_desc &restr, const pa_dense_basis_desc &basis, int qfunction,
                      const void *qf_ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops);
  void Finalize();  // operator.cpp:89-101
  // operator.cpp:103-114: the reference strips libCEED's assembly caches; this library keeps none between calls (the probing
  // workspace of pa_op_full_assemble and the diagonal's E-vector are released before those calls return), so there is nothing
  // to free -- kept so that callers (BilinearForm::Assemble, bilinearform.cpp:133-141) compile unchanged
  void DestroyAssemblyData() const;
  // operator.hpp:54; bilinearform.cpp:279 (discrete interpolators: the inverse multiplicity of the range dofs).  Mult /
  // AddMult scale the result, AddMultTranspose scales the input, as operator.cpp:181-240
  void SetDofMultiplicity(Vector &&mult);
  std::size_t Size() const;  // number of sub-operators (operator.hpp:46)
  const Context &GetContext() const { return *ctx_; }
  bool Streams() const { return pa_op_streams(op_) != 0; }  // y = A x runs on the streaming kernels
  void Mult(const Vector &x, Vector &y) const override;
  void MultTranspose(const Vector &x, Vector &y) const override;
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override;
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override;
  void AssembleDiagonal(Vector &diag) const override;
  bool IsSymmetric() const override;
  // multi-rank applies: the local dofs that take part in the halo exchange; MultAfter computes y = A x where those entries
  // of x are complete only once `after` has fired (interior element batches do not wait for it)
  void SetInterfaceDofs(const std::vector<int32_t> &ldofs);
  void MultAfter(const Vector &x, Vector &y, hipEvent_t after) const;
  // y = A (x with the essential entries read as zero), no copy of x (pa_op_mult_essential)
  void SetEssential(const int32_t *ess_host, int n);
  void MultEssential(const Vector &x, Vector &y) const;
  // the same + y[ess] = x[ess] | 0 inside the E^T kernels; returns false if the caller must fix the rows up
  bool MultEssentialDiag(const Vector &x, Vector &y, bool diag_one) const;
  // the Chebyshev step fused into E^T (pa_op_prepare_fused_step / pa_op_mult_cheb_step), essential list fused
  bool PrepareFusedStep() const;
  void MultChebyStepEssential(const Vector &x, const ChebyStepArgs &a, bool diag_one) const;
  void MultResidualEssential(const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0, Vector *d0, bool diag_one) const;
  // the fused forms on split vectors (pa_op_mult_split_step): interface dofs left to the halo kernel
  void MultSplitStep(const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *yg, int n_true,
                     int ess_policy, const pa_split_step &st) const;
  // split vectors (pa_op_mult_split): true dofs in x / y, ghosts read from xg0 | xg1 (parity of *sel) and written to yg
  bool SupportsSplit() const;
  void MultSplit(const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *y, double *yg,
                 int n_true, int ess_policy) const;
  // y = (Ar + i Ai) x in one pass over the element data (pa_op_mult_complex); ess_policy -1: plain, 0 / 1: with Ar's fused
  // essential list, rows set to 0 / x
  // 0: no such form; 1: tensor hexahedra (essential dofs can be fused); 2: dense tables (plain form only)
  static int ComplexFused(const Operator &Ar, const Operator &Ai) { return pa_op_complex_fused(Ar.op_, Ai.op_); }
  static void MultComplex(const Operator &Ar, const Operator &Ai, const Vector &xr, const Vector &xi, Vector &yr, Vector &yi,
                          int ess_policy = -1);
  // two right-hand sides in one pass over the element data (pa_op_mult2 / pa_op_mult2_essential_diag)
  void Mult2(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1) const;
  bool Mult2EssentialDiag(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1, bool diag_one) const;
};

}  // namespace ceed

// BaseSumOperator<Operator> (linalg/operator.hpp:132-270): y = sum_k a_k A_k x over operators of equal size
// (non-owning).  Palace's BuildParSumOperator (linalg/rap.cpp:843-919) wraps such a sum of local operators in
// one ParOperator to form a0 K + a1 C + a2 M.
class SumOperator : public Operator {
  const Context *ctx_;
  std::vector<std::pair<const Operator *, double>> ops_;
  mutable Vector z_;

public:
  SumOperator(const Context &ctx, int h, int w) : Operator(h, w), ctx_(&ctx) {}
  void AddOperator(const Operator &op, double a = 1.0);
  void Mult(const Vector &x, Vector &y) const override;
  void MultTranspose(const Vector &x, Vector &y) const override;
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override;
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override;
  void AssembleDiagonal(Vector &diag) const override;
  bool IsSymmetric() const override;
};

// BaseProductOperator<Operator> (linalg/operator.hpp:270-352): y = A (B x), non-owning
class ProductOperator : public Operator {
  const Operator &A_, &B_;
  mutable Vector z_;

public:
  ProductOperator(const Operator &A, const Operator &B) : Operator(A.Height(), B.Width()), A_(A), B_(B), z_(B.Height()) {}
  void Mult(const Vector &x, Vector &y) const override { B_.Mult(x, z_), A_.Mult(z_, y); }
  void MultTranspose(const Vector &x, Vector &y) const override {
    PA_REQUIRE(A_.Height() == A_.Width(), "the transposed product needs a square left factor (shared work vector)");
    A_.MultTranspose(x, z_), B_.MultTranspose(z_, y);
  }
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override { B_.Mult(x, z_), A_.AddMult(z_, y, a); }
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override {
    A_.MultTranspose(x, z_), B_.AddMultTranspose(z_, y, a);
  }
};

// BaseDiagonalOperator<Operator> (linalg/operator.hpp:354-423): y = d .* x, non-owning
class DiagonalOperator : public Operator {
  const Context *ctx_;
  const Vector &d_;

public:
  DiagonalOperator(const Context &ctx, const Vector &d) : Operator(d.Size(), d.Size()), ctx_(&ctx), d_(d) {}
  void Mult(const Vector &x, Vector &y) const override;
  void MultTranspose(const Vector &x, Vector &y) const override { Mult(x, y); }
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override;
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override { AddMult(x, y, a); }
  void AssembleDiagonal(Vector &diag) const override;
  bool IsSymmetric() const override { return true; }
};

// Local operator held as an assembled CSR matrix in device memory (csr_op.hip): what the reference's coarsest level
// becomes through ParOperator::ParallelAssemble (linalg/rap.cpp:84-152).  Non-owning view of a pa_csr.
class CsrOperator : public Operator {
  const Context *ctx_;
  const pa_csr *m_;
  int lanes_;
  void Apply(const double *vals, const Vector &x, Vector &y, double a, bool add) const;

public:
  CsrOperator(const Context &ctx, const pa_csr *m);
  ~CsrOperator() override;
  // one rank: ParOperator's essential-dof handling folded into a copy of the values (rows and columns zeroed, diagonal
  // 1 | 0).  The copy belongs to the caller (hipFree): several wrappers with different essential lists can share one matrix,
  // and this operator itself always applies the unconstrained values.
  double *EliminatedValues(const int32_t *d_ess, int n_ess, bool diag_one) const;
  void MultValues(const double *d_vals, const Vector &x, Vector &y) const;  // y = A' x, A' = this pattern with `d_vals`
  // the same on split vectors (pa_op_mult_split's contract: true dofs in x / y, ghosts read from xg0 | xg1 by the parity of *sel
  // and written to yg); d_vals == nullptr: the unconstrained values
  void MultSplit(const double *d_vals, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *y,
                 double *yg, int n_true) const;
  void Mult(const Vector &x, Vector &y) const override;
  void MultTranspose(const Vector &x, Vector &y) const override;  // symmetric matrices only (pa_csr::symmetric)
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override;
  void AssembleDiagonal(Vector &diag) const override;
  bool IsSymmetric() const override { return m_->symmetric; }
  const pa_csr &Matrix() const { return *m_; }
  // round 6: smoother steps / residuals in the epilogue of the sparse product (k_csr_spmv_step; PALACE_AMD_FUSED_STEP_CSR=0: off)
  bool PrepareChebyStep() const override;
  void MultChebyStep(const Vector &x, const ChebyStepArgs &a) const override;
  void MultResidual(const Vector &y, const Vector &b, Vector *res, const Vector *dinv = nullptr, double c0 = 0.0,
                    Vector *d0 = nullptr) const override;
  // the same with a caller's copy of the values (ParOperator's eliminated rows / columns: EliminatedValues)
  void MultChebyStepValues(const double *d_vals, const Vector &x, const ChebyStepArgs &a) const;
  void MultResidualValues(const double *d_vals, const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0, Vector *d0) const;
};

// ParOperator (rap.cpp:154-234): y = P^T A P x with essential-dof handling.  True dofs of this
// rank are the first n_true entries of the local (L-) vector; shared dofs owned elsewhere follow
// (see comm.hpp); with one rank P is the identity.
class ParOperator : public Operator {
public:
  enum class DiagonalPolicy { DIAG_ZERO = 0, DIAG_ONE = 1 };

private:
  const Context *ctx_;
  const Operator *A_;
  const ceed::Operator *A_fused_ = nullptr;  // single rank: BC masking fused into the local apply
  const ceed::Operator *A_overlap_ = nullptr;  // with a halo: interior elements run while the ghosts are exchanged
  const ceed::Operator *A_split_ = nullptr;    // peer transport: the local operator applies to split vectors (no L-vector copies)
  const ceed::Operator *A_split_avail_ = nullptr;
  const CsrOperator *A_csr_split_ = nullptr, *A_csr_split_avail_ = nullptr;  // the same for an assembled local operator
  bool split_ess_ = false;                     // ... with this wrapper's essential list fused into its index tables
  const CsrOperator *A_csr_ = nullptr;       // single rank, assembled local operator: BCs eliminated in the matrix values
  double *d_csr_bc_ = nullptr;               // ... this wrapper's copy of them (CsrOperator::EliminatedValues)
  const Halo *halo_;
  int n_true_, n_local_;
  int32_t *d_ess_ = nullptr;
  int n_ess_ = 0;
  std::vector<int32_t> ess_host_;
  uint8_t *d_ess_mask_ = nullptr;  // with a halo: one byte per true dof, so that copy + mask and copy + fix-up are one launch each
  DiagonalPolicy policy_;
  mutable Vector lx_, ly_;

public:
  ParOperator(const Context &ctx, const Operator &A, int n_true, const int32_t *ess_host, int n_ess,
              DiagonalPolicy policy, const Halo *halo = nullptr);
  ~ParOperator() override;
  const int32_t *GetEssentialTrueDofs() const { return d_ess_; }
  int NumEssentialTrueDofs() const { return n_ess_; }
  const std::vector<int32_t> &GetEssentialTrueDofsHost() const { return ess_host_; }
  const Operator &LocalOperator() const { return *A_; }
  // the direct form of the multi-rank Mult (no L-vector copies; peer transport + a local operator with a split-vector apply):
  // 1 in use, 0 available but switched off, -1 not available.  SetDirect(false) selects the L-vector form (A / B, verification).
  int DirectForm() const { return (A_split_ || A_csr_split_) ? 1 : ((A_split_avail_ || A_csr_split_avail_) ? 0 : -1); }
  void SetDirect(bool on) {
    A_split_ = on ? A_split_avail_ : nullptr, A_csr_split_ = on ? A_csr_split_avail_ : nullptr;
    StreamGraph::Invalidate();
  }
  bool FusesEssential() const { return A_fused_ != nullptr; }
  bool PrepareChebyStep() const override;
  void SplitStep(const Vector &x, const pa_split_step &st, const HaloStep &hs) const;
  void MultChebyStep(const Vector &x, const ChebyStepArgs &a) const override;
  void MultResidual(const Vector &y, const Vector &b, Vector *res, const Vector *dinv = nullptr, double c0 = 0.0,
                    Vector *d0 = nullptr) const override;  // the essential list lives in the local operator's index tables
  DiagonalPolicy GetDiagonalPolicy() const { return policy_; }
  const Halo *GetHalo() const { return halo_; }
  void Mult(const Vector &x, Vector &y) const override;
  // rap.cpp:236-275: y = P^T A^T P x with the same essential-dof handling (a symmetric local operator takes the fused
  // forward path)
  void MultTranspose(const Vector &x, Vector &y) const override;
  // y += a A x  (rap.cpp:277-318) / its transpose (:320-361)
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override;
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override;
  bool IsSymmetric() const override { return A_->IsSymmetric(); }
  // y0 = A x0, y1 = A x1: the real and imaginary parts of a complex vector through one real operator
  void Mult2(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1) const;
  // b -= A_unconstrained (x restricted to the essential dofs); b[ess] = x[ess] | 0  (rap.cpp:56-82)
  void EliminateRHS(const Vector &x, Vector &b) const;
  void AssembleDiagonal(Vector &diag) const override;

private:
  mutable Vector tt_;
};

// Solver<Operator> (solver.hpp:21-65)
class Solver : public Operator {
protected:
  bool initial_guess = false;

public:
  virtual void SetOperator(const Operator &op) = 0;
  // (not a re-configuration in the StreamGraph sense: composite solvers flip it inside their own recorded sequence)
  void SetInitialGuess(bool guess = true) { initial_guess = guess; }
  // y <- y + B (x - A y) style entry points used by the V-cycle (gmg.cpp:184,204)
  virtual void Mult2(const Vector &x, Vector &y, Vector &r) const;
  virtual void MultTranspose2(const Vector &x, Vector &y, Vector &r) const { Mult2(x, y, r); }
  // Surface failures a nested solver could only note on the device while it ran inside a recorded sequence (the inner PCG
  // of a multigrid cycle: non-finite (Br, r) / (Ap, p), the reference's CheckDot abort, iterative.cpp:39-45).  Outer Krylov
  // solvers call this on their preconditioner after a solve; composite solvers forward it to their parts.
  virtual void CheckStatus() const {}
};

namespace linalg {
double SpectralNorm(const Context &c, const Operator &A, const Vector &dinv, double tol = 1e-4, int max_it = 1000,
                    uint64_t seed = 0);
// batched inner products / updates of classical Gram-Schmidt (orthog.hpp:57-89): one pass, one all-reduce
void MultiDot(const Context &c, const Vector &w, const std::vector<Vector> &V, int m, double *H);
void MultiAXPY(const Context &c, const double *H, const std::vector<Vector> &V, int m, Vector &w);
}

class JacobiSmoother : public Solver {
  const Context *ctx_;
  const Operator *A_ = nullptr;
  Vector dinv_;

public:
  explicit JacobiSmoother(const Context &ctx) : ctx_(&ctx) {}
  void SetOperator(const Operator &op) override;
  void Mult(const Vector &x, Vector &y) const override;
};

class ChebyshevSmoother : public Solver {
  const Context *ctx_;
  int pc_it_, order_;
  double sf_max_, lambda_max_ = 0.0;
  bool fourth_kind_;
  double sf_min_;
  const Operator *A_ = nullptr;
  bool fused_step_ = false;  // A_ consumes A e_k inside its E^T (Operator::MultChebyStep)
  Vector dinv_;
  mutable Vector d_, r_, t_, w_;

public:
  ChebyshevSmoother(const Context &ctx, int smooth_it, int poly_order, double sf_max = 1.0, bool fourth_kind = true,
                    double sf_min = 0.0)
      : ctx_(&ctx), pc_it_(smooth_it), order_(poly_order), sf_max_(sf_max), fourth_kind_(fourth_kind),
        sf_min_(sf_min) {}
  void SetOperator(const Operator &op) override;
  double LambdaMax() const { return lambda_max_; }
  bool FusedStep() const { return fused_step_; }
  void Mult(const Vector &x, Vector &y) const override;
  void Mult2(const Vector &x, Vector &y, Vector &r) const override;
};

// Hiptmair distributive relaxation (distrelaxation.cpp:14-151): Chebyshev on the Nedelec operator,
// then Chebyshev on the auxiliary H1 operator through the discrete gradient G.
class DistRelaxationSmoother : public Solver {
  const Context *ctx_;
  int pc_it_;
  const Operator *G_;
  const Operator *A_ = nullptr;
  const ParOperator *A_G_ = nullptr;
  std::unique_ptr<ChebyshevSmoother> B_, B_G_;
  mutable Vector x_G_, y_G_, r_G_, t_;

public:
  DistRelaxationSmoother(const Context &ctx, const Operator &G, int smooth_it, int cheby_smooth_it, int cheby_order,
                         double cheby_sf_max = 1.0, double cheby_sf_min = 0.0, bool cheby_4th_kind = true);
  void SetOperator(const Operator &) override { throw pa::Error("use SetOperators(op, op_G)"); }
  void SetOperators(const Operator &op, const ParOperator &op_G);
  const ChebyshevSmoother &Primary() const { return *B_; }
  const ChebyshevSmoother &Auxiliary() const { return *B_G_; }
  void Mult(const Vector &x, Vector &y) const override;
  void Mult2(const Vector &x, Vector &y, Vector &r) const override;
  void MultTranspose2(const Vector &x, Vector &y, Vector &r) const override;
};

// Iterative solver base (iterative.hpp:25-115)
class IterativeSolver : public Solver {
protected:
  const Context *ctx_;
  const Operator *A_ = nullptr;
  const Solver *B_ = nullptr;
  double rel_tol_ = 0.0, abs_tol_ = 0.0;
  int max_it_ = 100;
  mutable bool converged_ = false;
  mutable double initial_res_ = 1.0, final_res_ = 0.0;
  mutable int final_it_ = 0;
  int print_ = 0;

public:
  explicit IterativeSolver(const Context &ctx, int print = 0) : ctx_(&ctx), print_(print) {}
  void SetOperator(const Operator &op) override {
    A_ = &op, height = op.Height(), width = op.Width();
    StreamGraph::Invalidate();
  }
  virtual void SetPreconditioner(const Solver &pc) { B_ = &pc, StreamGraph::Invalidate(); }
  void SetTol(double tol) { rel_tol_ = tol, StreamGraph::Invalidate(); }
  void SetAbsTol(double tol) { abs_tol_ = tol, StreamGraph::Invalidate(); }
  void SetMaxIter(int its) { max_it_ = its, StreamGraph::Invalidate(); }
  void CheckStatus() const override {
    Finish();
    if (B_) B_->CheckStatus();
  }
  bool GetConverged() const { return Finish(), converged_; }
  double GetInitialRes() const { return Finish(), initial_res_; }
  double GetFinalRes() const { return Finish(), final_res_; }
  int GetNumIterations() const { return Finish(), final_it_; }

protected:
  // solvers that leave their statistics on the device until someone asks (CgSolver with device-resident scalars)
  virtual void Finish() const {}
};

// CgSolver (iterative.cpp:360-486).  Default form: the scalars of the recurrence (beta, (Ap, p), alpha, the
// residual norm, the convergence decision) live on the device and every kernel reads them there, so an iteration is
// a fixed launch sequence with no host round trip: one fused dot + (multi-rank) one all-reduce per inner product,
// the iteration recorded as a HIP graph, and the host running `lookahead` iterations ahead of the last residual it
// has seen.  Once the device decides the solve has converged the remaining enqueued iterations change nothing
// (every update kernel returns at once), so iterates, iteration counts and residuals are those of the reference's
// check-every-iteration loop.  SetHostScalars(true) (or PALACE_AMD_CG_HOST=1) gives that loop literally.
class CgSolver : public IterativeSolver {
  mutable Vector r_, z_, p_;
  struct DeviceState;
  mutable std::unique_ptr<DeviceState> dev_;
  int lookahead_ = 1;
  bool host_scalars_ = false;
  void MultHost(const Vector &b, Vector &x) const;
  void MultDevice(const Vector &b, Vector &x) const;
  void Finish() const override;

public:
  explicit CgSolver(const Context &ctx, int print = 0);
  ~CgSolver() override;
  void Mult(const Vector &b, Vector &x) const override;
  // iterations the host may enqueue beyond the last one whose residual it has read back; < 0: never wait (inner
  // solvers with a small iteration cap: their statistics stay on the device until asked for)
  void SetLookahead(int k) { lookahead_ = k; }
  void SetHostScalars(bool host) { host_scalars_ = host; }
  void SetOperator(const Operator &op) override;
  void SetPreconditioner(const Solver &pc) override;
};

enum class Orthogonalization { MGS = 0, CGS = 1, CGS2 = 2 };  // config "Orthogonalization", orthog.hpp:41-89

namespace linalg {
// OrthogonalizeColumnMGS / OrthogonalizeColumnCGS (linalg/orthog.hpp:41-89): H[j] = (w, V[j]) for j < m and
// w -= sum_j H[j] V[j]; the inputs are assumed normalised, the output is not normalised.  `weight` (optional, square)
// makes the inner product (w, v) = v^T W w as the reference's weighted inner-product helpers do
// (test/unit/test-orthog.cpp:21-68).  CGS2 = CGS with one refinement pass (refine = true).
void OrthogonalizeColumn(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, double *H,
                         int m, const Operator *weight = nullptr);
// The three statements of a GMRES step (iterative.cpp:629-633): orthogonalise w against V[0 .. m), return H(m, .) = ||w|| and
// normalise w -- with the coefficients kept on the device between the kernels (orthog.hip): one host synchronisation per column.
// PALACE_AMD_GS=host runs the three calls one after the other with the host in between (the form of rounds 1-4).
double OrthonormalizeColumn(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, double *H, int m);
// (device form of OrthogonalizeColumn: inner products of `x` when given -- W w of a weighted inner product --, else of w)
void OrthogonalizeColumnDevice(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, const Vector *x,
                               double *H, int m, bool normalize, double *hn);
bool DeviceOrthogonalization();
long long ResidentColumns();  // columns orthogonalised with w resident in registers so far (orthog.hip: k_mgs_resident)
void SetDeviceOrthogonalization(bool on);  // A / B switch (default on; PALACE_AMD_GS=host starts with it off)
}  // namespace linalg

enum class PreconditionerSide { LEFT = 0, RIGHT = 1 };  // config "PCSide" (iterative.hpp:187-214)

// GmresSolver (iterative.cpp:543-705): restarted GMRES, left (default) or right preconditioning; flexible = true is
// FgmresSolver (iterative.cpp:733-871): right preconditioning with the preconditioned basis stored.  One implementation
// for real and complex scalars (krylov_impl.hpp).
class GmresSolver : public IterativeSolver {
protected:
  int max_dim_ = -1;
  bool flexible_ = false;
  PreconditionerSide pc_side_ = PreconditionerSide::LEFT;
  Orthogonalization orthog_ = Orthogonalization::MGS;
  mutable std::vector<Vector> V_, Z_;
  mutable Vector r_;

public:
  GmresSolver(const Context &ctx, int print = 0, bool flexible = false)
      : IterativeSolver(ctx, print), flexible_(flexible), pc_side_(flexible ? PreconditionerSide::RIGHT : PreconditionerSide::LEFT) {}
  void SetRestartDim(int dim) { max_dim_ = dim; }
  void SetOrthogonalization(Orthogonalization o) { orthog_ = o; }
  virtual void SetPreconditionerSide(PreconditionerSide side) {
    PA_REQUIRE(!flexible_ || side == PreconditionerSide::RIGHT, "FGMRES solver only supports right preconditioning!");  // iterative.hpp:268-272
    pc_side_ = side;
  }
  void Mult(const Vector &b, Vector &x) const override;
};

class FgmresSolver : public GmresSolver {
public:
  explicit FgmresSolver(const Context &ctx, int print = 0) : GmresSolver(ctx, print, true) {}
};

// A problem small enough to be solved redundantly by every rank -- the coarsest multigrid level, where the reference runs
// HYPRE's distributed AMS / BoomerAMG (linalg/ams.cpp, amg.cpp; ksp.cpp:143-157) and the native cycles of amg_solver.hpp work on
// one rank's matrix: the right-hand side, distributed as T-vectors, is gathered into a vector in GLOBAL numbering on every rank
// (a halo plan over that vector: each rank sends its true dofs, receives everybody else's), every rank applies the same solver
// to the same global vector -- same matrix, same numbering, hence the same bits everywhere -- and keeps its own entries.  The
// coarse problem is 1 / 27 of the fine one at p = 3; replicating it costs no communication besides the gather.
class ReplicatedSolver : public Solver {
  const Context *ctx_;
  const Halo *gather_;    // plan on the global-numbered vector: send = my true dofs (global numbers), recv = the other ranks'
  const Solver *inner_;   // solver of the global problem (n_global x n_global)
  int n_true_, n_global_;
  int32_t *d_mine_ = nullptr;  // [n_true] global number of my true dof i
  double *d_sign_ = nullptr;   // [n_true] +-1: orientation of my dof relative to the global one (nullptr: all +1)
  mutable Vector gx_, gy_;

public:
  // sign_host (optional): dof i of this rank is sign[i] times global dof mine[i] -- rank-local meshes may orient an edge against
  // the global numbering
  ReplicatedSolver(const Context &ctx, const Halo &gather, const Solver &inner, const int32_t *mine_host, int n_true, int n_global,
                   const double *sign_host = nullptr);
  ~ReplicatedSolver() override;
  void SetOperator(const Operator &) override {}  // the global problem is the inner solver's
  void Mult(const Vector &x, Vector &y) const override;
  void CheckStatus() const override { inner_->CheckStatus(); }
};

// GeometricMultigridSolver (gmg.cpp): levels 0 (coarsest) .. L-1, prolongations P[l]: level l -> l+1
class GeometricMultigridSolver : public Solver {
  const Context *ctx_;
  int pc_it_;
  std::vector<const Operator *> P_;
  std::vector<const ParOperator *> A_;
  std::vector<std::unique_ptr<Solver>> B_;
  mutable std::vector<Vector> X_, Y_, R_;  // (the finest X_ / Y_ are views of the caller's vectors: Mult)
  mutable Vector Xown_, Yown_;              // the finest vectors of applications to varying (x, y)
  mutable StreamGraph graph_, graph_alias_;  // one application (pc_it V-cycles): on the solver's vectors / on a caller's recurring pair
  mutable const double *last_x_ = nullptr, *last_y_ = nullptr;
  std::vector<char> fused_res_;  // level l: r = x - A y comes out of the operator's E^T epilogue (Operator::MultResidual)
  void VCycle(int l, bool initial_guess) const;

public:
  // G (optional): discrete gradients per level => DistRelaxationSmoother on levels >= 1 (gmg.cpp:41-60)
  GeometricMultigridSolver(const Context &ctx, std::unique_ptr<Solver> &&coarse_solver,
                           const std::vector<const Operator *> &P, int cycle_it, int smooth_it, int cheby_order,
                           double cheby_sf_max = 1.0, double cheby_sf_min = 0.0, bool cheby_4th_kind = true,
                           const std::vector<const Operator *> *G = nullptr);
  // Operators for every level (the reference passes a MultigridOperator, gmg.cpp:69-123); aux_ops are
  // the auxiliary-space (H1) operators required when G was given
  void SetOperators(const std::vector<const ParOperator *> &ops,
                    const std::vector<const ParOperator *> *aux_ops = nullptr);
  void SetOperator(const Operator &) override { throw pa::Error("use SetOperators for multigrid"); }
  void Mult(const Vector &x, Vector &y) const override;
  const Solver &Smoother(int l) const { return *B_[l]; }
  void CheckStatus() const override {
    for (const auto &b : B_) b->CheckStatus();
  }
};

}  // namespace palace
