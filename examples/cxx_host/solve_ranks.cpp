// C++ host example, several ranks: one process per rank (per GPU; several per GPU work too), no MPI, no Python at run time.
// Each rank reads what Palace's ParMesh / ParFiniteElementSpace / GroupCommunicator would hold for it (dump_problem_ranks.py: its
// z-slab of the cavity, the spaces of every multigrid level in local numbering with the true dofs first, the halo plans), brings up
// the peer transport of comm.hpp (arena handles exchanged through files in a shared directory -- MPI_Allgather in Palace), and
// solves (K + eps_r M) x = b with the classes of fem.hpp / ksp.hpp exactly as solve.cpp does on one rank:
//   KspSolver: PCG + p-multigrid with Hiptmair smoothing and LinearSolver::AMS on the lowest-order level.
// With a halo on that level AMS is the ReplicatedCoarseSolver (ksp.hpp): the global level-0 problem is assembled from the ranks'
// pieces by the C++ layer and solved redundantly -- where the reference hands HYPRE the distributed matrix (linalg/ksp.cpp:129-239).
//
// coarse = amg (round 5): the H1 problem of BASELINE config 4's shape instead -- (eps grad u, grad v) = b on the H1 spaces, PCG +
// p-multigrid with LinearSolver::BOOMER_AMG on the lowest-order level.  With a halo that is the native V-cycle with its solve
// DISTRIBUTED over the ranks (amg_dist.hpp; every rank its rows of every algebraic level, one halo exchange per product);
// PALACE_AMD_COARSE_SOLVE=replicated: the whole cycle on the gathered problem on every rank (rounds 3-4).
//
//   ./solve_ranks prefix rank world dir [coarse=ams|pcg|amg]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "ksp.hpp"

using namespace palace;

static std::vector<std::vector<char>> read_blobs(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw pa::Error("cannot open " + path);
  int64_t n = 0;
  f.read(reinterpret_cast<char *>(&n), 8);
  std::vector<std::vector<char>> out((size_t)n);
  for (auto &b : out) {
    int64_t bytes = 0;
    f.read(reinterpret_cast<char *>(&bytes), 8);
    b.resize((size_t)bytes);
    f.read(b.data(), bytes);
  }
  return out;
}

// the 64-byte arena handles of all ranks through files (rank r writes dir/handle.r, reads the others')
static std::vector<char> exchange_handles(Comm &comm, const std::string &dir, int rank, int world) {
  char mine[Comm::kPeerHandleBytes];
  comm.PeerHandle(mine);
  {
    const std::string tmp = dir + "/handle." + std::to_string(rank) + ".tmp", fin = dir + "/handle." + std::to_string(rank);
    std::ofstream f(tmp, std::ios::binary);
    f.write(mine, sizeof(mine));
    f.close();
    std::rename(tmp.c_str(), fin.c_str());
  }
  std::vector<char> all((size_t)world * Comm::kPeerHandleBytes);
  for (int r = 0; r < world; r++) {
    const std::string path = dir + "/handle." + std::to_string(r);
    for (int tries = 0;; tries++) {
      std::ifstream f(path, std::ios::binary);
      if (f && f.read(&all[(size_t)r * Comm::kPeerHandleBytes], Comm::kPeerHandleBytes)) break;
      if (tries > 6000) throw pa::Error("no arena handle from rank " + std::to_string(r));
      std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
  }
  return all;
}

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const std::string prefix = argv[1], dir = argv[4], coarse = argc > 5 ? argv[5] : "ams";
  const int rank = std::atoi(argv[2]), world = std::atoi(argv[3]);
  try {
    auto blobs = read_blobs(prefix + "." + std::to_string(rank));
    auto i32 = [&](size_t i) { return reinterpret_cast<const int32_t *>(blobs[i].data()); };
    auto f64 = [&](size_t i) { return reinterpret_cast<const double *>(blobs[i].data()); };
    const int ne = i32(0)[0], nn = i32(0)[1], order = i32(0)[2], nlev = i32(0)[3];

    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) throw pa::Error("no HIP device");
    Comm comm(rank, world);  // peer transport only
    if (world > 1) {
      const std::vector<char> handles = exchange_handles(comm, dir, rank, world);
      comm.PeerConnect(handles.data());
    }
    Context ctx;
    ctx.stream = stream, ctx.comm = &comm;

    fem::DefaultIntegrationOrder::p_trial = order;
    Mesh mesh(ctx, ne, 2, nn, i32(1), f64(2), i32(3), fem::DefaultIntegrationOrder::GetQ1d(2 * order - 1));

    std::vector<std::unique_ptr<Halo>> halos;
    auto make_halo = [&](size_t b) -> const Halo * {
      if (world == 1) return nullptr;
      const int nnbr = (int)(blobs[b].size() / 4);
      halos.push_back(std::make_unique<Halo>(comm, nnbr, i32(b), i32(b + 1), i32(b + 2), i32(b + 3), i32(b + 4)));
      return halos.back().get();
    };
    FiniteElementSpaceHierarchy nd_fespaces, h1_fespaces;
    for (int l = 0; l < nlev; l++) {
      const size_t b = 4 + 17 * (size_t)l;
      const int p = i32(0)[4 + l], nd_size = i32(b)[0], h1_size = i32(b)[1], nd_true = i32(b)[2], h1_true = i32(b)[3];
      const Halo *nd_halo = make_halo(b + 7), *h1_halo = make_halo(b + 12);  // (every rank creates its plans in the same order)
      auto nd = std::make_unique<FiniteElementSpace>(ctx, mesh, PA_FE_HCURL, p, nd_size, i32(b + 1),
                                                     reinterpret_cast<const uint8_t *>(blobs[b + 2].data()), i32(b + 3),
                                                     world > 1 ? nd_true : -1, nd_halo);
      nd->SetEssentialTrueDofs(i32(b + 4), (int)(blobs[b + 4].size() / 4));
      nd_fespaces.AddLevel(std::move(nd));
      auto h1 = std::make_unique<FiniteElementSpace>(ctx, mesh, PA_FE_H1, p, h1_size, i32(b + 5), nullptr, nullptr,
                                                     world > 1 ? h1_true : -1, h1_halo);
      h1->SetEssentialTrueDofs(i32(b + 6), (int)(blobs[b + 6].size() / 4));
      h1_fespaces.AddLevel(std::move(h1));
    }

    MaterialPropertyCoefficient muinv(1), eps(1);
    muinv.AddMaterialProperty(1, 1.0);
    eps.AddMaterialProperty(1, 2.08);
    BilinearForm::pa_order_threshold = 2;  // the p = 1 level becomes a matrix (the reference assembles it for AMS)
    if (coarse == "amg") {  // the scalar problem: H1 spaces only
      BilinearForm g(h1_fespaces.GetFinestFESpace());
      g.AddDomainIntegrator<DiffusionIntegrator>(eps);
      auto g_ops = g.Assemble(h1_fespaces, false);
      auto A = std::make_unique<MultigridOperator>(h1_fespaces.GetNumLevels());
      for (std::size_t l = 0; l < h1_fespaces.GetNumLevels(); l++) {
        const auto &fes = h1_fespaces.GetFESpaceAtLevel(l);
        auto op = std::make_unique<FespaceParOperator>(std::move(g_ops[l]), fes);
        op->SetEssentialTrueDofs(fes.GetEssentialTrueDofs(), ParOperator::DiagonalPolicy::DIAG_ONE);
        A->AddOperator(std::move(op));
      }
      config::LinearSolverData linear;
      linear.krylov_solver = KrylovSolver::CG;
      linear.type = LinearSolver::BOOMER_AMG;
      linear.tol = 1e-10, linear.max_it = 400;
      linear.initial_guess = 0;
      linear.SetDefaults(order, /*spd_problem=*/true);
      KspSolver ksp(linear, /*verbose=*/0, h1_fespaces);
      ksp.SetOperators(*A, *A);
      const int n = A->Height();
      Vector ones(n), rhs(n), x(n), res(n);
      linalg::Fill(ctx, ones, 1.0);
      {  // b = M 1: the load vector of a constant source (the same function whatever the partition)
        BilinearForm m(h1_fespaces.GetFinestFESpace());
        m.AddDomainIntegrator<MassIntegrator>(eps);
        auto m_ops = m.Assemble(h1_fespaces, false);
        FespaceParOperator M(std::move(m_ops.back()), h1_fespaces.GetFinestFESpace());
        M.Mult(ones, rhs);
      }
      const auto &ess = A->GetFinestOperator().Par();
      linalg::SetSubVector(ctx, rhs, ess.GetEssentialTrueDofs(), ess.NumEssentialTrueDofs(), 0.0);
      ksp.Mult(rhs, x);
      A->Mult(x, res);
      linalg::AXPBY(ctx, 1.0, rhs, -1.0, res);
      const double rn = linalg::Norml2(ctx, res) / linalg::Norml2(ctx, rhs), sx = linalg::Dot(ctx, x, ones), nglob = linalg::Dot(ctx, ones, ones);
      comm.PeerCheck(stream);
      if (rank == 0)
        std::printf("cxx_host_ranks: world %d  order %d  levels %d  global ndofs %d  coarse %s  iterations %d  converged %d  "
                    "|b - A x| / |b| %.3e  sum(x) %.12e\n",
                    world, order, nlev, (int)std::lround(nglob), coarse.c_str(), ksp.GetKrylovSolver().GetNumIterations(),
                    (int)ksp.GetKrylovSolver().GetConverged(), rn, sx);
      if (world > 1) comm.Barrier(stream);
      return 0;
    }
    BilinearForm a(nd_fespaces.GetFinestFESpace());
    a.AddDomainIntegrator<CurlCurlMassIntegrator>(muinv, eps);
    auto a_ops = a.Assemble(nd_fespaces, /*skip_zeros=*/false);
    auto A = std::make_unique<MultigridOperator>(nd_fespaces.GetNumLevels());
    for (std::size_t l = 0; l < nd_fespaces.GetNumLevels(); l++) {
      const auto &fes = nd_fespaces.GetFESpaceAtLevel(l);
      auto op = std::make_unique<FespaceParOperator>(std::move(a_ops[l]), fes);
      op->SetEssentialTrueDofs(fes.GetEssentialTrueDofs(), ParOperator::DiagonalPolicy::DIAG_ONE);
      A->AddOperator(std::move(op));
    }
    BilinearForm g(h1_fespaces.GetFinestFESpace());
    g.AddDomainIntegrator<DiffusionIntegrator>(eps);
    auto g_ops = g.Assemble(h1_fespaces, false);
    for (std::size_t l = 0; l < h1_fespaces.GetNumLevels(); l++) {
      const auto &fes = h1_fespaces.GetFESpaceAtLevel(l);
      auto op = std::make_unique<FespaceParOperator>(std::move(g_ops[l]), fes);
      op->SetEssentialTrueDofs(fes.GetEssentialTrueDofs(), ParOperator::DiagonalPolicy::DIAG_ONE);
      A->AddAuxiliaryOperator(std::move(op));
    }

    config::LinearSolverData linear;
    linear.krylov_solver = KrylovSolver::CG;
    linear.type = coarse == "pcg" ? LinearSolver::JACOBI_PCG : LinearSolver::AMS;
    linear.tol = 1e-10, linear.max_it = 400;
    linear.mg_smooth_aux = 1;
    linear.initial_guess = 0;
    linear.SetDefaults(order, /*spd_problem=*/true);
    linear.mg_smooth_aux = 1;
    KspSolver ksp(linear, /*verbose=*/0, nd_fespaces, &h1_fespaces);
    ksp.SetOperators(*A, *A);

    const int n = A->Height();
    Vector ones(n), rhs(n), x(n), res(n);
    linalg::Fill(ctx, ones, 1.0);
    A->Mult(ones, rhs);
    const auto &ess = A->GetFinestOperator().Par();
    linalg::SetSubVector(ctx, rhs, ess.GetEssentialTrueDofs(), ess.NumEssentialTrueDofs(), 0.0);
    ksp.Mult(rhs, x);
    A->Mult(x, res);
    linalg::AXPBY(ctx, 1.0, rhs, -1.0, res);
    const double rn = linalg::Norml2(ctx, res) / linalg::Norml2(ctx, rhs), sx = linalg::Dot(ctx, x, ones), nglob = linalg::Dot(ctx, ones, ones);
    comm.PeerCheck(stream);
    if (rank == 0)
      std::printf("cxx_host_ranks: world %d  order %d  levels %d  global ndofs %d  coarse %s  iterations %d  converged %d  "
                  "|b - A x| / |b| %.3e  sum(x) %.12e\n",
                  world, order, nlev, (int)std::lround(nglob), coarse.c_str(), ksp.GetKrylovSolver().GetNumIterations(),
                  (int)ksp.GetKrylovSolver().GetConverged(), rn, sx);
    if (world > 1) comm.Barrier(stream);  // nobody unmaps an arena a neighbour may still store into
  } catch (const std::exception &e) {
    std::fprintf(stderr, "palace_amd (rank %d): %s\n", rank, e.what());  // MFEM_ABORT in Palace
    return 1;
  }
  return 0;
}
