// KspSolver (see ksp.hpp).  Host code only.
#include "ksp.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "amg_dist.hpp"
#include "amg_solver.hpp"

namespace palace {

namespace {

// a few Jacobi-preconditioned CG iterations as one Solver (stand-in coarse solve); never waits for the host
class JacobiPcgSolver : public Solver {
  JacobiSmoother jac_;
  CgSolver cg_;

public:
  JacobiPcgSolver(const Context &ctx, double tol, int max_it) : jac_(ctx), cg_(ctx) {
    cg_.SetTol(tol), cg_.SetMaxIter(max_it), cg_.SetLookahead(-1);
    cg_.SetPreconditioner(jac_);
  }
  void SetOperator(const Operator &op) override {
    height = op.Height(), width = op.Width();
    jac_.SetOperator(op), cg_.SetOperator(op);
  }
  void Mult(const Vector &x, Vector &y) const override {
    const_cast<CgSolver &>(cg_).SetInitialGuess(initial_guess);
    cg_.Mult(x, y);
  }
  void CheckStatus() const override { cg_.CheckStatus(); }
};

// ---- native algebraic coarse solvers (amg_solver.hpp) behind the reference's configuration names ----------------------
// The reference hands the assembled coarse operator to HYPRE (BoomerAmgSolver, linalg/amg.cpp; HypreAmsSolver, linalg/ams.cpp).
// HYPRE is an external package outside /root/reference; LinearSolver::BOOMER_AMG / AMS select the smoothed-aggregation
// V-cycle / the auxiliary-space Maxwell cycle of amg_solver.hip instead (same role, same inputs: the assembled matrix of
// the level, and for AMS the lowest-order discrete gradient and the vertex coordinates, ams.cpp:64-100).  One rank only.
amg::HostCsr AssembledLevelMatrix(const Operator &op, std::vector<char> &ess_flag) {
  const auto *fop = dynamic_cast<const FespaceParOperator *>(&op);
  const ParOperator *par = fop ? &fop->Par() : dynamic_cast<const ParOperator *>(&op);
  PA_REQUIRE(par, "the algebraic coarse solvers need a ParOperator (the level operator of the multigrid hierarchy)");
  PA_REQUIRE(!par->GetHalo(), "the native AMG / AMS coarse solvers run on one rank (multi-rank: JACOBI_PCG or "
                              "CHEBYSHEV_JACOBI)");
  const auto &ess = par->GetEssentialTrueDofsHost();
  ess_flag.assign((size_t)op.Height(), 0);
  for (const int32_t d : ess) ess_flag[d] = 1;
  if (const auto *csr = dynamic_cast<const CsrOperator *>(&par->LocalOperator()))
    return DownloadCsr(csr->Matrix(), ess.data(), (int)ess.size());
  const auto *pa = dynamic_cast<const ceed::Operator *>(&par->LocalOperator());
  PA_REQUIRE(pa, "the coarse operator is neither assembled nor a partially assembled operator");
  const auto m = BilinearForm::FullAssemble(*pa, /*skip_zeros=*/false);  // rap.cpp:84-152 does this for HYPRE
  return DownloadCsr(m->Matrix(), ess.data(), (int)ess.size());
}

// ---- the coarsest level across ranks: replicated (ksp.hpp: ReplicatedCoarseSolver) ------------------------------------------
// in-place sum over the ranks of a host array (Mpi::GlobalSum on a std::vector)
void GlobalSumHost(const Context &ctx, std::vector<double> &v) {
  if (!ctx.comm || ctx.comm->Size() == 1 || v.empty()) return;
  double *d = pa::dev_upload(v.data(), v.size(), ctx.stream);
  ctx.comm->AllReduceSum(d, (int)v.size(), ctx.stream);
  PA_HIP(hipMemcpyAsync(v.data(), d, sizeof(double) * v.size(), hipMemcpyDeviceToHost, ctx.stream));
  PA_HIP(hipStreamSynchronize(ctx.stream));
  (void)hipFree(d);
  ctx.comm->PeerCheckNow();
}
// the pieces of all ranks one after the other (rank order); counts / offsets of the pieces on request
std::vector<double> AllGatherV(const Context &ctx, const std::vector<double> &mine, std::vector<long long> *offsets = nullptr) {
  const int size = ctx.comm ? ctx.comm->Size() : 1, rank = ctx.comm ? ctx.comm->Rank() : 0;
  // peer transport: a real gather (every value crosses once per reader); RCCL-only and in-process communicators: as a sum of
  // zero-padded global arrays (below)
  if (ctx.comm && size > 1 && ctx.comm->PeerReady() && !std::getenv("PALACE_AMD_GATHER_BY_SUM")) {
    PA_HIP(hipStreamSynchronize(ctx.stream));
    return ctx.comm->AllGatherVHost(mine, offsets);
  }
  std::vector<double> cnt((size_t)size, 0.0);
  cnt[(size_t)rank] = (double)mine.size();
  GlobalSumHost(ctx, cnt);
  std::vector<long long> off((size_t)size + 1, 0);
  for (int r = 0; r < size; r++) off[(size_t)r + 1] = off[(size_t)r] + (long long)cnt[(size_t)r];
  PA_REQUIRE(off[(size_t)size] < (1ll << 31), "replicated coarse level: too many entries to gather");
  std::vector<double> all((size_t)off[(size_t)size], 0.0);
  std::copy(mine.begin(), mine.end(), all.begin() + off[(size_t)rank]);
  GlobalSumHost(ctx, all);  // (zeros outside a rank's own segment: the sum is the concatenation, exactly)
  if (offsets) *offsets = off;
  return all;
}
// local CSR of the level operator in L-vector numbering, nothing eliminated
amg::HostCsr LocalLevelMatrix(const ParOperator &par) {
  if (const auto *csr = dynamic_cast<const CsrOperator *>(&par.LocalOperator())) return DownloadCsr(csr->Matrix(), nullptr, 0);
  const auto *pa = dynamic_cast<const ceed::Operator *>(&par.LocalOperator());
  PA_REQUIRE(pa, "the coarse operator is neither assembled nor a partially assembled operator");
  const auto m = BilinearForm::FullAssemble(*pa, /*skip_zeros=*/false);
  return DownloadCsr(m->Matrix(), nullptr, 0);
}

}  // namespace

struct ReplicatedCoarseSolver::Impl {
  std::unique_ptr<Solver> dist;  // distributed solve (round 5: DistAmgSolver / DistAmsSolver): owned pieces in, owned pieces out
  std::unique_ptr<Solver> inner;  // AmsSolver / AmgSolver of the global problem
  std::unique_ptr<Halo> gather;
  std::unique_ptr<ReplicatedSolver> rep;
  int n_global = 0;
};
ReplicatedCoarseSolver::~ReplicatedCoarseSolver() = default;
int ReplicatedCoarseSolver::GlobalSize() const { return impl_->n_global; }
bool ReplicatedCoarseSolver::Distributed() const { return impl_->dist != nullptr; }
const Solver *ReplicatedCoarseSolver::DistributedSolver() const { return impl_->dist.get(); }
void ReplicatedCoarseSolver::Mult(const Vector &x, Vector &y) const {
  if (impl_->dist) return impl_->dist->Mult(x, y);
  impl_->rep->Mult(x, y);
}

ReplicatedCoarseSolver::ReplicatedCoarseSolver(const Context &ctx, const Operator &level0, const Operator *G, int nv_true,
                                               const double *xyz_true, int dim, int cycle_it, bool singular)
    : impl_(new Impl) {
  StreamGraph::RequireNotRecording("ReplicatedCoarseSolver set-up");
  const auto *fop = dynamic_cast<const FespaceParOperator *>(&level0);
  const ParOperator *par = fop ? &fop->Par() : dynamic_cast<const ParOperator *>(&level0);
  PA_REQUIRE(par && ctx.comm, "replicated coarse solver: needs the level's ParOperator and a communicator");
  const Halo *halo = par->GetHalo();
  PA_REQUIRE(halo, "replicated coarse solver: the level has no halo plan (one rank: use the solver directly)");
  const int size = ctx.comm->Size(), rank = ctx.comm->Rank();
  const int n_true = level0.Height(), n_local = par->LocalOperator().Height();
  height = width = n_true;
  // ---- global numbers: true dofs rank by rank, ghosts through the halo plan
  std::vector<long long> off;
  {
    std::vector<double> one((size_t)n_true, 1.0);  // (only the piece sizes are wanted)
    std::vector<double> cnt((size_t)size, 0.0);
    cnt[(size_t)rank] = (double)n_true;
    GlobalSumHost(ctx, cnt);
    off.assign((size_t)size + 1, 0);
    for (int r = 0; r < size; r++) off[(size_t)r + 1] = off[(size_t)r] + (long long)cnt[(size_t)r];
  }
  const long long n_global = off[(size_t)size];
  PA_REQUIRE(n_global < (1ll << 31), "replicated coarse level: too many global dofs");
  impl_->n_global = (int)n_global;
  std::vector<double> gid((size_t)n_local, -1.0);
  for (int i = 0; i < n_true; i++) gid[(size_t)i] = (double)(off[(size_t)rank] + i);
  {
    Vector lx(n_local);
    PA_HIP(hipMemcpyAsync(lx.Data(), gid.data(), sizeof(double) * (size_t)n_local, hipMemcpyHostToDevice, ctx.stream));
    halo->Prolongate(lx.Data(), ctx.stream);
    PA_HIP(hipMemcpyAsync(gid.data(), lx.Data(), sizeof(double) * (size_t)n_local, hipMemcpyDeviceToHost, ctx.stream));
    PA_HIP(hipStreamSynchronize(ctx.stream));
    ctx.comm->PeerCheckNow();
  }
  for (int i = 0; i < n_local; i++) PA_REQUIRE(gid[(size_t)i] >= 0.0 && gid[(size_t)i] < (double)n_global, "replicated coarse level: a ghost has no owner");
  // ---- the global matrix: triplets of every rank's local matrix in global numbers
  amg::HostCsr Ag;
  {
    const amg::HostCsr Al = LocalLevelMatrix(*par);
    PA_REQUIRE(Al.nrows == n_local && Al.ncols == n_local, "replicated coarse level: local matrix size");
    std::vector<double> ti((size_t)Al.nnz()), tj((size_t)Al.nnz());
    for (int r = 0; r < n_local; r++)
      for (int a = Al.rowptr[(size_t)r]; a < Al.rowptr[(size_t)r + 1]; a++) ti[(size_t)a] = gid[(size_t)r], tj[(size_t)a] = gid[(size_t)Al.col[(size_t)a]];
    const std::vector<double> gi = AllGatherV(ctx, ti), gj = AllGatherV(ctx, tj), gv = AllGatherV(ctx, Al.val);
    const size_t nt = gi.size();
    // rows by counting (entries of a row stay in gathered order: rank by rank), then columns sorted, duplicates summed in that
    // order -- the same arithmetic on every rank
    std::vector<int> rp((size_t)n_global + 1, 0);
    for (size_t k = 0; k < nt; k++) rp[(size_t)gi[k] + 1]++;
    for (long long r = 0; r < n_global; r++) rp[(size_t)r + 1] += rp[(size_t)r];
    std::vector<int> pos(rp.begin(), rp.end() - 1), ord(nt);
    for (size_t k = 0; k < nt; k++) ord[(size_t)pos[(size_t)gi[k]]++] = (int)k;
    Ag.nrows = Ag.ncols = (int)n_global;
    Ag.rowptr.assign((size_t)n_global + 1, 0);
    for (long long r = 0; r < n_global; r++) {
      auto b = ord.begin() + rp[(size_t)r], e = ord.begin() + rp[(size_t)r + 1];
      std::stable_sort(b, e, [&](int x, int y) { return gj[(size_t)x] < gj[(size_t)y]; });
      for (auto it = b; it != e; ++it) {
        const int c = (int)gj[(size_t)*it];
        if (!Ag.col.empty() && (int)Ag.col.size() > Ag.rowptr[(size_t)r] && Ag.col.back() == c)
          Ag.val.back() += gv[(size_t)*it];
        else
          Ag.col.push_back(c), Ag.val.push_back(gv[(size_t)*it]);
      }
      Ag.rowptr[(size_t)r + 1] = (int)Ag.col.size();
    }
  }
  // ---- essential dofs (global flags), eliminated from the matrix like ParOperator::ParallelAssemble does (rap.cpp:131-149)
  std::vector<char> ess_flag((size_t)n_global, 0);
  {
    const auto &ess = par->GetEssentialTrueDofsHost();
    std::vector<double> eg(ess.size());
    for (size_t i = 0; i < ess.size(); i++) eg[i] = gid[(size_t)ess[i]];
    for (const double g : AllGatherV(ctx, eg)) ess_flag[(size_t)g] = 1;
    amg::HostCsr e;
    e.nrows = e.ncols = Ag.nrows;
    e.rowptr.assign((size_t)Ag.nrows + 1, 0);
    for (int r = 0; r < Ag.nrows; r++) {
      if (ess_flag[(size_t)r]) {
        e.col.push_back(r), e.val.push_back(1.0);
      } else {
        for (int a = Ag.rowptr[(size_t)r]; a < Ag.rowptr[(size_t)r + 1]; a++)
          if (!ess_flag[(size_t)Ag.col[(size_t)a]]) e.col.push_back(Ag.col[(size_t)a]), e.val.push_back(Ag.val[(size_t)a]);
      }
      e.rowptr[(size_t)r + 1] = (int)e.col.size();
    }
    Ag = std::move(e);
  }
  // The SOLVE distributed over the ranks (amg_dist.hpp: every rank keeps and applies its rows of every level of the algebraic
  // hierarchies, one owner -> ghost exchange per product) unless PALACE_AMD_COARSE_SOLVE=replicated (read at construction): the
  // whole cycle on the gathered problem on every rank, rounds 3-4
  const char *mode = std::getenv("PALACE_AMD_COARSE_SOLVE");
  bool distributed = !(mode && std::string(mode) == "replicated");
  {
    // the mode is read per process: agree on it before branching (a rank with a different environment would skip the collective
    // halo constructions of the distributed form and the others would wait for it until the peer time-out) -- replicated if ANY
    // rank asks for it
    std::vector<double> votes((size_t)size, 0.0);
    votes[(size_t)rank] = distributed ? 0.0 : 1.0;
    GlobalSumHost(ctx, votes);
    for (const double v : votes) distributed = distributed && v == 0.0;
  }
  const std::vector<int> ioff(off.begin(), off.end());
  if (G) {
    // ---- the lowest-order discrete gradient in global numbers: two applications of the (multi-rank) operator to the global
    // vertex numbers and their squares identify both ends of every true edge of this rank; rows gathered in rank order are
    // the rows of the global matrix
    PA_REQUIRE(G->Height() == n_true && G->Width() == nv_true && xyz_true, "replicated AMS: gradient / coordinates do not match the level");
    std::vector<long long> voff;
    std::vector<double> xyz_all;
    {
      std::vector<double> mine(xyz_true, xyz_true + (size_t)nv_true * dim);
      xyz_all = AllGatherV(ctx, mine, &voff);
    }
    const long long nv_global = voff[(size_t)size] / dim;
    PA_REQUIRE((double)nv_global * (double)nv_global < 9.0e15, "too many vertices for the two-probe reconstruction of the gradient");
    const long long v0 = voff[(size_t)rank] / dim;
    std::vector<double> x1((size_t)nv_true), x2((size_t)nv_true), d((size_t)n_true), q((size_t)n_true);
    for (int v = 0; v < nv_true; v++) x1[(size_t)v] = (double)(v0 + v) + 1.0, x2[(size_t)v] = x1[(size_t)v] * x1[(size_t)v];
    Vector dx(nv_true), dy(n_true);
    auto apply = [&](const std::vector<double> &in, std::vector<double> &out) {
      PA_HIP(hipMemcpyAsync(dx.Data(), in.data(), sizeof(double) * (size_t)nv_true, hipMemcpyHostToDevice, ctx.stream));
      G->Mult(dx, dy);
      PA_HIP(hipMemcpyAsync(out.data(), dy.Data(), sizeof(double) * (size_t)n_true, hipMemcpyDeviceToHost, ctx.stream));
      PA_HIP(hipStreamSynchronize(ctx.stream));
      ctx.comm->PeerCheckNow();
    };
    apply(x1, d), apply(x2, q);
    std::vector<double> head((size_t)n_true), tail((size_t)n_true);
    for (int e = 0; e < n_true; e++) {
      const double diff = std::round(d[(size_t)e]);
      PA_REQUIRE(diff != 0.0 && std::abs(d[(size_t)e] - diff) < 1e-6, "the discrete gradient is not an edge-vertex incidence matrix "
                                                                        "(AMS needs the lowest-order spaces on level 0)");
      const double sum = std::round(q[(size_t)e] / diff);
      head[(size_t)e] = (sum + diff) / 2 - 1, tail[(size_t)e] = (sum - diff) / 2 - 1;
    }
    const std::vector<double> gh = AllGatherV(ctx, head), gt = AllGatherV(ctx, tail);
    PA_REQUIRE((long long)gh.size() == n_global, "replicated AMS: gathered gradient rows");
    amg::HostCsr Gm;
    Gm.nrows = (int)n_global, Gm.ncols = (int)nv_global;
    Gm.rowptr.resize((size_t)n_global + 1);
    Gm.col.resize((size_t)2 * n_global), Gm.val.resize((size_t)2 * n_global);
    for (long long e = 0; e < n_global; e++) {
      const int h = (int)gh[(size_t)e], t = (int)gt[(size_t)e];
      PA_REQUIRE(h >= 0 && h < nv_global && t >= 0 && t < nv_global && h != t, "gradient reconstruction failed");
      Gm.rowptr[(size_t)e] = (int)(2 * e);
      const bool hf = h < t;
      Gm.col[(size_t)(2 * e)] = hf ? h : t, Gm.val[(size_t)(2 * e)] = hf ? 1.0 : -1.0;
      Gm.col[(size_t)(2 * e + 1)] = hf ? t : h, Gm.val[(size_t)(2 * e + 1)] = hf ? -1.0 : 1.0;
    }
    Gm.rowptr[(size_t)n_global] = (int)(2 * n_global);
    AmsOptions opt;
    opt.cycle_it = std::max(cycle_it, 1), opt.singular = singular;
    if (distributed) {
      std::vector<int> ivoff((size_t)size + 1);
      for (int r = 0; r <= size; r++) ivoff[(size_t)r] = (int)(voff[(size_t)r] / dim);
      auto ams = std::make_unique<DistAmsSolver>(ctx, Ag, Gm, xyz_all.data(), dim, ess_flag, ioff, ivoff, opt);
      if (std::getenv("PALACE_AMD_COARSE_VERBOSE")) {
        for (const DistAmgSolver *amg : {ams->GradientSpaceSolver(), ams->NodalSpaceSolver()}) {
          if (!amg) continue;
          std::string msg = "palace_amd: distributed AMS, rank " + std::to_string(rank) +
                            (amg == ams->NodalSpaceSolver() ? ", nodal spaces:" : ", gradient space:");
          for (int l = 0; l < amg->NumLevels(); l++)
            msg += " [" + std::to_string(amg->LevelRows(l)) + " rows, " + std::to_string(amg->LevelOwned(l)) + " owned + " +
                   std::to_string(amg->LevelGhosts(l)) + " ghosts]";
          std::fprintf(stderr, "%s\n", msg.c_str());
        }
      }
      impl_->dist = std::move(ams);
      return;
    }
    impl_->inner = std::make_unique<AmsSolver>(ctx, Ag, Gm, xyz_all.data(), dim, ess_flag, opt);
  } else {
    if (distributed) {
      auto amg = std::make_unique<DistAmgSolver>(ctx, Ag, ioff);
      if (std::getenv("PALACE_AMD_COARSE_VERBOSE")) {  // the hierarchy as this rank holds it
        std::string msg = "palace_amd: distributed AMG, rank " + std::to_string(rank) + ":";
        for (int l = 0; l < amg->NumLevels(); l++)
          msg += " [" + std::to_string(amg->LevelRows(l)) + " rows, " + std::to_string(amg->LevelOwned(l)) + " owned + " +
                 std::to_string(amg->LevelGhosts(l)) + " ghosts]";
        std::fprintf(stderr, "%s\n", msg.c_str());
      }
      impl_->dist = std::move(amg);
      return;
    }
    impl_->inner = std::make_unique<AmgSolver>(ctx, Ag);
  }
  // ---- the gather plan on the global-numbered vector: my true dofs to everybody, everybody else's pieces (contiguous) in
  std::vector<int> nbr, soff(1, 0), roff(1, 0);
  std::vector<int32_t> sidx, ridx, mine((size_t)n_true);
  for (int i = 0; i < n_true; i++) mine[(size_t)i] = (int32_t)(off[(size_t)rank] + i);
  for (int r = 0; r < size; r++) {
    if (r == rank) continue;
    nbr.push_back(r);
    sidx.insert(sidx.end(), mine.begin(), mine.end());
    soff.push_back((int)sidx.size());
    for (long long g = off[(size_t)r]; g < off[(size_t)r + 1]; g++) ridx.push_back((int32_t)g);
    roff.push_back((int)ridx.size());
  }
  impl_->gather = std::make_unique<Halo>(*ctx.comm, (int)nbr.size(), nbr.data(), soff.data(), sidx.data(), roff.data(), ridx.data());
  impl_->rep = std::make_unique<ReplicatedSolver>(ctx, *impl_->gather, *impl_->inner, mine.data(), n_true, (int)n_global);
}

namespace {

class NativeAmgSolver : public Solver {  // LinearSolver::BOOMER_AMG
  const Context *ctx_;
  int cycle_it_;
  std::unique_ptr<AmgSolver> amg_;
  std::unique_ptr<ReplicatedCoarseSolver> rep_;  // several ranks: the global level solved redundantly
  mutable Vector r_, z_;

public:
  NativeAmgSolver(const Context &ctx, int cycle_it) : ctx_(&ctx), cycle_it_(std::max(cycle_it, 1)) {}
  void SetOperator(const Operator &op) override {
    StreamGraph::Invalidate();
    height = op.Height(), width = op.Width();
    A_ = &op;
    const auto *fop = dynamic_cast<const FespaceParOperator *>(&op);
    const ParOperator *par = fop ? &fop->Par() : dynamic_cast<const ParOperator *>(&op);
    if (par && par->GetHalo()) {
      amg_.reset();
      rep_ = std::make_unique<ReplicatedCoarseSolver>(*ctx_, op, nullptr, 0, nullptr, 3, 1, false);
      return;
    }
    rep_.reset();
    std::vector<char> ess_flag;
    amg_ = std::make_unique<AmgSolver>(*ctx_, AssembledLevelMatrix(op, ess_flag));
  }
  void Mult(const Vector &b, Vector &x) const override {
    if (rep_) {
      // (further cycles as stationary iterations, as below)
      if (!initial_guess) rep_->Mult(b, x);
      for (int it = initial_guess ? 0 : 1; it < cycle_it_; it++) {
        r_.SetSize(height), z_.SetSize(height);
        A_->Mult(x, r_);
        linalg::AXPBY(*ctx_, 1.0, b, -1.0, r_);
        rep_->Mult(r_, z_);
        linalg::AXPY(*ctx_, 1.0, z_, x);
      }
      return;
    }
    PA_REQUIRE(amg_, "NativeAmgSolver: SetOperator first");
    // the cycle itself starts from zero; a caller's guess (Solver::SetInitialGuess) enters as the first stationary step
    if (!initial_guess) amg_->Mult(b, x);
    for (int it = initial_guess ? 0 : 1; it < cycle_it_; it++) {  // (further) cycles as stationary iterations x += B (b - A x)
      r_.SetSize(height), z_.SetSize(height);
      A_->Mult(x, r_);
      linalg::AXPBY(*ctx_, 1.0, b, -1.0, r_);
      amg_->Mult(r_, z_);
      linalg::AXPY(*ctx_, 1.0, z_, x);
    }
  }

private:
  const Operator *A_ = nullptr;
};

class NativeAmsSolver : public Solver {  // LinearSolver::AMS
  const Context *ctx_;
  const FiniteElementSpace *nd_, *h1_;
  AmsOptions opt_;
  std::unique_ptr<AmsSolver> ams_;
  std::unique_ptr<ReplicatedCoarseSolver> rep_;  // several ranks: the global level solved redundantly
  const Operator *A_ = nullptr;
  mutable Vector r_, z_;

  // the lowest-order discrete gradient as a matrix: every row is +1 at the edge's head and -1 at its tail, so two applications
  // (to the vertex numbers and to their squares) identify both vertices of every edge
  amg::HostCsr GradientMatrix() const {
    const Operator &G = nd_->GetDiscreteInterpolator(*h1_);
    const int nv = h1_->GetTrueVSize(), ne = nd_->GetTrueVSize();
    PA_REQUIRE((double)nv * nv < 9.0e15, "too many vertices for the two-probe reconstruction of the gradient");
    std::vector<double> x1((size_t)nv), x2((size_t)nv), d((size_t)ne), q((size_t)ne);
    for (int v = 0; v < nv; v++) x1[v] = v + 1.0, x2[v] = (v + 1.0) * (v + 1.0);
    Vector dx(nv), dy(ne);
    auto apply = [&](const std::vector<double> &in, std::vector<double> &out) {
      PA_HIP(hipMemcpyAsync(dx.Data(), in.data(), sizeof(double) * (size_t)nv, hipMemcpyHostToDevice, ctx_->stream));
      G.Mult(dx, dy);
      PA_HIP(hipMemcpyAsync(out.data(), dy.Data(), sizeof(double) * (size_t)ne, hipMemcpyDeviceToHost, ctx_->stream));
      PA_HIP(hipStreamSynchronize(ctx_->stream));
      if (ctx_->comm) ctx_->comm->PeerCheckNow();
    };
    apply(x1, d), apply(x2, q);
    amg::HostCsr Gm;
    Gm.nrows = ne, Gm.ncols = nv;
    Gm.rowptr.resize((size_t)ne + 1);
    Gm.col.resize((size_t)2 * ne), Gm.val.resize((size_t)2 * ne);
    for (int e = 0; e < ne; e++) {
      const double diff = std::round(d[e]);
      PA_REQUIRE(diff != 0.0 && std::abs(d[e] - diff) < 1e-6, "the discrete gradient is not an edge-vertex incidence matrix "
                                                              "(AMS needs the lowest-order spaces on level 0)");
      const double sum = std::round(q[e] / diff);
      const int head = (int)((sum + diff) / 2) - 1, tail = (int)((sum - diff) / 2) - 1;
      PA_REQUIRE(head >= 0 && head < nv && tail >= 0 && tail < nv && head != tail, "gradient reconstruction failed");
      Gm.rowptr[e] = 2 * e;
      const bool head_first = head < tail;
      Gm.col[2 * e] = head_first ? head : tail, Gm.val[2 * e] = head_first ? 1.0 : -1.0;
      Gm.col[2 * e + 1] = head_first ? tail : head, Gm.val[2 * e + 1] = head_first ? -1.0 : 1.0;
    }
    Gm.rowptr[ne] = 2 * ne;
    return Gm;
  }

public:
  NativeAmsSolver(const Context &ctx, const FiniteElementSpace &nd, const FiniteElementSpace &h1, int cycle_it, int singular)
      : ctx_(&ctx), nd_(&nd), h1_(&h1) {
    PA_REQUIRE(nd.GetMaxElementOrder() == 1 && h1.GetMaxElementOrder() == 1,
               "the native AMS solver is built for the lowest-order level (the reference's default coarse level); use it as "
               "the coarse solver of a p-multigrid hierarchy");
    opt_.cycle_it = std::max(cycle_it, 1);
    opt_.singular = singular > 0;
  }
  void SetOperator(const Operator &op) override {
    StreamGraph::Invalidate();
    height = op.Height(), width = op.Width();
    A_ = &op;
    if (nd_->GetHalo()) {  // several ranks (ksp.cpp:129-239 hands HYPRE the distributed matrix; here: gathered, solved everywhere)
      ams_.reset();
      const std::vector<double> xyz = nd_->GetMesh().VertexCoordinates(*h1_);  // (true vertices first: the first GetTrueVSize() rows)
      rep_ = std::make_unique<ReplicatedCoarseSolver>(*ctx_, op, &nd_->GetDiscreteInterpolator(*h1_), h1_->GetTrueVSize(), xyz.data(),
                                                      nd_->GetMesh().SpaceDimension(), opt_.cycle_it, opt_.singular);
      return;
    }
    rep_.reset();
    std::vector<char> ess_flag;
    const amg::HostCsr A = AssembledLevelMatrix(op, ess_flag);
    const amg::HostCsr G = GradientMatrix();
    const std::vector<double> xyz = nd_->GetMesh().VertexCoordinates(*h1_);
    ams_ = std::make_unique<AmsSolver>(*ctx_, A, G, xyz.data(), nd_->GetMesh().SpaceDimension(), ess_flag, opt_);
  }
  void Mult(const Vector &b, Vector &x) const override {
    if (rep_) {  // several ranks: the replicated solve starts from zero; a caller's guess enters as one residual correction
      if (!initial_guess) return rep_->Mult(b, x);  // (the same rule as NativeAmgSolver above, and as on one rank)
      r_.SetSize(height), z_.SetSize(height);
      A_->Mult(x, r_);
      linalg::AXPBY(*ctx_, 1.0, b, -1.0, r_);
      rep_->Mult(r_, z_);
      linalg::AXPY(*ctx_, 1.0, z_, x);
      return;
    }
    PA_REQUIRE(ams_, "NativeAmsSolver: SetOperator first");
    ams_->SetInitialGuess(initial_guess);  // (the cycles of AmsSolver::Mult honour it in their first smoothing step)
    ams_->Mult(b, x);
  }
};

std::unique_ptr<IterativeSolver> ConfigureKrylovSolver(const config::LinearSolverData &linear, int verbose,
                                                       const Context &ctx) {
  // ksp.cpp:27-106
  std::unique_ptr<IterativeSolver> ksp;
  const auto type = linear.krylov_solver;
  switch (type) {
    case KrylovSolver::CG:
      ksp = std::make_unique<CgSolver>(ctx, verbose);
      break;
    case KrylovSolver::GMRES: {
      auto gmres = std::make_unique<GmresSolver>(ctx, verbose);
      gmres->SetRestartDim(linear.max_size);
      ksp = std::move(gmres);
    } break;
    case KrylovSolver::FGMRES: {
      auto fgmres = std::make_unique<FgmresSolver>(ctx, verbose);
      fgmres->SetRestartDim(linear.max_size);
      ksp = std::move(fgmres);
    } break;
    default:
      throw pa::Error("Unexpected solver type for Krylov solver configuration!");
  }
  ksp->SetInitialGuess(linear.initial_guess > 0);
  ksp->SetTol(linear.tol);
  ksp->SetMaxIter(linear.max_it);
  if (linear.pc_side != PreconditionerSideOption::DEFAULT && type != KrylovSolver::GMRES) {
    std::fprintf(stderr, "Warning: Preconditioner side will be ignored for non-GMRES iterative solvers!\n");
  } else if (type == KrylovSolver::GMRES || type == KrylovSolver::FGMRES) {
    auto *gmres = static_cast<GmresSolver *>(ksp.get());
    if (linear.pc_side == PreconditionerSideOption::LEFT) gmres->SetPreconditionerSide(PreconditionerSide::LEFT);
    if (linear.pc_side == PreconditionerSideOption::RIGHT) gmres->SetPreconditionerSide(PreconditionerSide::RIGHT);
  }
  if (type == KrylovSolver::GMRES || type == KrylovSolver::FGMRES)
    static_cast<GmresSolver *>(ksp.get())->SetOrthogonalization(linear.gs_orthog);
  return ksp;
}

std::unique_ptr<Solver> ConfigurePreconditionerSolver(const config::LinearSolverData &linear, int verbose, const Context &ctx,
                                                      const FiniteElementSpaceHierarchy &fespaces,
                                                      const FiniteElementSpaceHierarchy *aux_fespaces) {
  // ksp.cpp:131-258: the solver of the coarsest level (or of the only level), then the multigrid hierarchy around it
  std::unique_ptr<Solver> pc;
  switch (linear.type) {
    case LinearSolver::JACOBI:
      pc = std::make_unique<JacobiSmoother>(ctx);
      break;
    case LinearSolver::CHEBYSHEV_JACOBI:
      pc = std::make_unique<ChebyshevSmoother>(ctx, 1, linear.coarse_order);
      break;
    case LinearSolver::JACOBI_PCG:
      pc = std::make_unique<JacobiPcgSolver>(ctx, linear.coarse_tol, linear.coarse_max_it);
      break;
    case LinearSolver::AMS: {
      // ksp.cpp:143-152: the coarse solve of the multigrid hierarchy or the solver of the only level
      PA_REQUIRE(aux_fespaces, "AMS solver relies on both primary space and auxiliary spaces for construction!");
      const bool coarse_solver = fespaces.GetNumLevels() > 1;
      pc = std::make_unique<NativeAmsSolver>(ctx, fespaces.GetFESpaceAtLevel(0), aux_fespaces->GetFESpaceAtLevel(0),
                                             coarse_solver ? linear.ams_max_it : linear.mg_cycle_it, linear.ams_singular_op);
    } break;
    case LinearSolver::BOOMER_AMG:
      pc = std::make_unique<NativeAmgSolver>(ctx, fespaces.GetNumLevels() > 1 ? 1 : linear.mg_cycle_it);  // ksp.cpp:153-157
      break;
    case LinearSolver::MUMPS:
    case LinearSolver::SUPERLU:
    case LinearSolver::STRUMPACK:
    case LinearSolver::STRUMPACK_MP:
    case LinearSolver::CUDSS:
      throw pa::Error("the sparse direct solvers live in external packages and are not part of palace_amd: use AMS, "
                      "BOOMER_AMG (native algebraic cycles), JACOBI, CHEBYSHEV_JACOBI or JACOBI_PCG");
    default:
      throw pa::Error("Unexpected solver type for preconditioner configuration!");
  }
  if (fespaces.GetNumLevels() > 1) {
    const auto P = fespaces.GetProlongationOperators();
    const int order = linear.mg_smooth_order > 0 ? linear.mg_smooth_order
                                                 : std::max(2 * fespaces.GetFinestFESpace().GetMaxElementOrder(), 4);
    if (linear.mg_smooth_aux > 0) {
      PA_REQUIRE(aux_fespaces, "Multigrid with auxiliary space smoothers requires both primary space and auxiliary spaces "
                               "for construction!");
      const auto G = fespaces.GetDiscreteInterpolators(*aux_fespaces);
      return std::make_unique<GeometricMultigridSolver>(ctx, std::move(pc), P, std::max(linear.mg_cycle_it, 1),
                                                        linear.mg_smooth_it, order, linear.mg_smooth_sf_max,
                                                        linear.mg_smooth_sf_min, linear.mg_smooth_cheby_4th, &G);
    }
    return std::make_unique<GeometricMultigridSolver>(ctx, std::move(pc), P, std::max(linear.mg_cycle_it, 1),
                                                      linear.mg_smooth_it, order, linear.mg_smooth_sf_max,
                                                      linear.mg_smooth_sf_min, linear.mg_smooth_cheby_4th, nullptr);
  }
  (void)verbose;
  return pc;
}

}  // namespace

void config::LinearSolverData::SetDefaults(int order, bool spd_problem) {
  if (krylov_solver == KrylovSolver::DEFAULT) krylov_solver = spd_problem ? KrylovSolver::CG : KrylovSolver::GMRES;
  if (type == LinearSolver::DEFAULT) type = LinearSolver::JACOBI_PCG;  // (reference: AMS / BoomerAMG / a direct solver)
  if (ams_max_it < 0) ams_max_it = 1;       // iodata.cpp: one AMS cycle per coarse solve
  if (ams_singular_op < 0) ams_singular_op = 0;
  if (max_size < 0) max_size = max_it;
  if (initial_guess < 0) initial_guess = 1;
  if (mg_max_levels < 0) mg_max_levels = 100;
  if (mg_cycle_it < 0) mg_cycle_it = 1;
  if (mg_smooth_aux < 0) mg_smooth_aux = spd_problem ? 0 : 1;
  if (mg_smooth_order < 0) mg_smooth_order = std::max(2 * order, 4);
}

std::vector<int> GetPolynomialOrders(int order, MultigridCoarsening coarsening, int mg_max_levels) {
  // multigrid.hpp:44-69: LINEAR p -> p - 1, LOGARITHMIC p -> (p + 1) / 2, down to 1
  std::vector<int> out{order};
  while (out.back() > 1 && (mg_max_levels < 0 || (int)out.size() < mg_max_levels))
    out.push_back(coarsening == MultigridCoarsening::LINEAR ? out.back() - 1 : (out.back() + 1) / 2);
  std::reverse(out.begin(), out.end());
  return out;
}

KspSolver::KspSolver(const config::LinearSolverData &linear, int verbose, const FiniteElementSpaceHierarchy &fespaces,
                     const FiniteElementSpaceHierarchy *aux_fespaces)
    : KspSolver(ConfigureKrylovSolver(linear, verbose, fespaces.GetFinestFESpace().GetContext()),
                ConfigurePreconditionerSolver(linear, verbose - 1, fespaces.GetFinestFESpace().GetContext(), fespaces,
                                              aux_fespaces)) {}

KspSolver::KspSolver(std::unique_ptr<IterativeSolver> &&ksp_, std::unique_ptr<Solver> &&pc_)
    : ksp(std::move(ksp_)), pc(std::move(pc_)) {
  if (pc) ksp->SetPreconditioner(*pc);
}

namespace {
void SetPreconditionerOperators(Solver &pc_ref, const Operator &pc_op) {
  Solver *pc = &pc_ref;
  const auto *mg_op = dynamic_cast<const MultigridOperator *>(&pc_op);
  auto *mg_pc = dynamic_cast<GeometricMultigridSolver *>(pc);
  if (mg_pc) {
    PA_REQUIRE(mg_op, "GeometricMultigridSolver requires a MultigridOperator argument provided to SetOperator!");
    std::vector<const ParOperator *> ops, aux;
    for (std::size_t l = 0; l < mg_op->GetNumLevels(); l++) ops.push_back(&mg_op->GetOperatorAtLevel(l).Par());
    for (std::size_t l = 0; l < mg_op->GetNumAuxiliaryLevels(); l++) aux.push_back(&mg_op->GetAuxiliaryOperatorAtLevel(l).Par());
    mg_pc->SetOperators(ops, mg_op->HasAuxiliaryOperators() ? &aux : nullptr);
  } else if (mg_op) {
    pc->SetOperator(mg_op->GetFinestOperator());
  } else {
    pc->SetOperator(pc_op);
  }
}
}  // namespace

void KspSolver::SetOperators(const Operator &op, const Operator &pc_op) {
  // ksp.cpp:295-313; a multigrid preconditioner takes the operators of all levels (gmg.cpp:69-123)
  PhaseRange range("Linear Solve / Setup");  // ksp.cpp:297
  ksp->SetOperator(op);
  if (pc) SetPreconditionerOperators(*pc, pc_op);
}

void KspSolver::Mult(const Vector &x, Vector &y) const {
  PhaseRange range("Linear Solve");  // ksp.cpp:317
  ksp->Mult(x, y);
  if (!ksp->GetConverged())
    std::fprintf(stderr, "Warning: Linear solver did not converge, norm(Ax-b)/norm(b) = %.3e (norm(b) = %.3e)!\n",
                 ksp->GetFinalRes() / ksp->GetInitialRes(), ksp->GetInitialRes());
  ksp_mult++;
  ksp_mult_it += ksp->GetNumIterations();
}

ComplexKspSolver::ComplexKspSolver(const config::LinearSolverData &linear, int verbose,
                                   const FiniteElementSpaceHierarchy &fespaces,
                                   const FiniteElementSpaceHierarchy *aux_fespaces) {
  const Context &ctx = fespaces.GetFinestFESpace().GetContext();
  // ksp.cpp:27-106 for OperType = ComplexOperator
  switch (linear.krylov_solver) {
    case KrylovSolver::CG:
      ksp = std::make_unique<ComplexCgSolver>(ctx, verbose);
      if (linear.pc_side != PreconditionerSideOption::DEFAULT)
        std::fprintf(stderr, "Warning: Preconditioner side will be ignored for non-GMRES iterative solvers!\n");
      break;
    case KrylovSolver::GMRES:
    case KrylovSolver::FGMRES: {
      const bool flexible = linear.krylov_solver == KrylovSolver::FGMRES;
      auto gmres = std::make_unique<ComplexGmresSolver>(ctx, verbose, flexible);
      gmres->SetRestartDim(linear.max_size);
      gmres->SetOrthogonalization(linear.gs_orthog);
      if (!flexible && linear.pc_side == PreconditionerSideOption::RIGHT) gmres->SetPreconditionerSide(PreconditionerSide::RIGHT);
      if (!flexible && linear.pc_side == PreconditionerSideOption::LEFT) gmres->SetPreconditionerSide(PreconditionerSide::LEFT);
      ksp = std::move(gmres);
    } break;
    default:
      throw pa::Error("Unexpected solver type for Krylov solver configuration!");
  }
  ksp->SetTol(linear.tol), ksp->SetMaxIter(linear.max_it);
  initial_guess = linear.initial_guess > 0;
  pc = ConfigurePreconditionerSolver(linear, verbose - 1, ctx, fespaces, aux_fespaces);
  ksp->SetPreconditioner(*pc);
}

void ComplexKspSolver::SetOperators(const ComplexOperator &op, const Operator &pc_op) {
  PhaseRange range("Linear Solve / Setup");
  ksp->SetOperator(op);
  SetPreconditionerOperators(*pc, pc_op);
}

void ComplexKspSolver::Mult(const ComplexVector &x, ComplexVector &y) const {
  PhaseRange range("Linear Solve");
  ksp->Mult(x, y, initial_guess);
  if (!ksp->GetConverged())
    std::fprintf(stderr, "Warning: Linear solver did not converge, norm(Ax-b)/norm(b) = %.3e (norm(b) = %.3e)!\n",
                 ksp->GetFinalRes() / ksp->GetInitialRes(), ksp->GetInitialRes());
  ksp_mult++;
  ksp_mult_it += ksp->GetNumIterations();
}

}  // namespace palace
