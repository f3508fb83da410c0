// C++ host example with the reference's full multigrid hierarchy: h-levels (the lowest order of the p-sequence on every mesh of
// a uniform-refinement sequence) under the p-levels of the finest mesh -- ConstructFiniteElementSpaceHierarchy,
// fem/multigrid.hpp:77-123.  From the arrays of dump_problem_hp.py: one Mesh per refinement level with its
// GetRefinementTransforms() data, FiniteElementSpaceHierarchy::AddLevel across the meshes, BilinearForm::Assemble(hierarchy)
// (a new partial assembly on every mesh, p-coarsened copies on one mesh: bilinearform.cpp:153-201), the refinement transfer
// between meshes built by the hierarchy (fespace.cpp:246-251), KspSolver (PCG + V-cycle over all levels).
//   ./solve_hp problem.bin [aux=0|1] [coarse=pcg|ams|cheb]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "ksp.hpp"

using namespace palace;

static std::vector<std::vector<char>> read_blobs(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    std::fprintf(stderr, "cannot open %s\n", path);
    std::exit(2);
  }
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

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  const bool aux = argc > 2 && std::atoi(argv[2]) != 0;
  const std::string coarse = argc > 3 ? argv[3] : "pcg";
  try {
    auto blobs = read_blobs(argv[1]);
    auto i32 = [&](size_t i) { return reinterpret_cast<const int32_t *>(blobs[i].data()); };
    auto f64 = [&](size_t i) { return reinterpret_cast<const double *>(blobs[i].data()); };
    const int nmesh = i32(0)[0], order = i32(0)[1], nlev = i32(0)[2];
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) throw pa::Error("no HIP device");
    Context ctx;
    ctx.stream = stream;
    fem::DefaultIntegrationOrder::p_trial = order;  // the quadrature follows the solution order on every level and mesh

    std::vector<std::unique_ptr<Mesh>> mesh;  // geodata.cpp:426-460: every mesh of the refinement sequence is kept
    size_t b = 1;
    for (int m = 0; m < nmesh; m++) {
      const int ne = i32(b)[0], nn = i32(b)[1];
      mesh.push_back(std::make_unique<Mesh>(ctx, ne, 2, nn, i32(b + 1), f64(b + 2), i32(b + 3), fem::DefaultIntegrationOrder::GetQ1d(2 * 3 - 1)));
      b += 4;
      if (m > 0) {
        mesh[m]->SetRefinementTransforms(*mesh[m - 1], i32(b), i32(b + 1), 8, f64(b + 2));
        b += 3;
      }
    }
    FiniteElementSpaceHierarchy nd_fespaces, h1_fespaces;
    for (int l = 0; l < nlev; l++, b += 7) {
      const int m = i32(0)[3 + 2 * l], p = i32(0)[4 + 2 * l], nd_size = i32(b)[0], h1_size = i32(b)[1];
      auto nd = std::make_unique<FiniteElementSpace>(ctx, *mesh[m], PA_FE_HCURL, p, nd_size, i32(b + 1),
                                                     reinterpret_cast<const uint8_t *>(blobs[b + 2].data()), i32(b + 3));
      nd->SetEssentialTrueDofs(i32(b + 4), (int)(blobs[b + 4].size() / 4));
      nd_fespaces.AddLevel(std::move(nd));
      auto h1 = std::make_unique<FiniteElementSpace>(ctx, *mesh[m], PA_FE_H1, p, h1_size, i32(b + 5), nullptr, nullptr);
      h1->SetEssentialTrueDofs(i32(b + 6), (int)(blobs[b + 6].size() / 4));
      h1_fespaces.AddLevel(std::move(h1));
    }
    MaterialPropertyCoefficient muinv(1), eps(1);
    muinv.AddMaterialProperty(1, 1.0);
    eps.AddMaterialProperty(1, 2.08);
    BilinearForm::pa_order_threshold = 1;  // (KspSolver assembles the coarsest level itself where the coarse solver needs a matrix)
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
    if (aux) {
      BilinearForm g(h1_fespaces.GetFinestFESpace());
      g.AddDomainIntegrator<DiffusionIntegrator>(eps);
      auto g_ops = g.Assemble(h1_fespaces, false);
      for (std::size_t l = 0; l < h1_fespaces.GetNumLevels(); l++) {
        const auto &fes = h1_fespaces.GetFESpaceAtLevel(l);
        auto op = std::make_unique<FespaceParOperator>(std::move(g_ops[l]), fes);
        op->SetEssentialTrueDofs(fes.GetEssentialTrueDofs(), ParOperator::DiagonalPolicy::DIAG_ONE);
        A->AddAuxiliaryOperator(std::move(op));
      }
    }
    config::LinearSolverData linear;
    linear.krylov_solver = KrylovSolver::CG;
    linear.type = coarse == "pcg" ? LinearSolver::JACOBI_PCG : coarse == "ams" ? LinearSolver::AMS : LinearSolver::CHEBYSHEV_JACOBI;
    linear.tol = 1e-10, linear.max_it = 400;
    linear.mg_smooth_aux = aux ? 1 : 0;
    linear.initial_guess = 0;
    linear.SetDefaults(order, /*spd_problem=*/true);
    KspSolver ksp(linear, /*verbose=*/0, nd_fespaces, (aux || coarse == "ams") ? &h1_fespaces : nullptr);
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
    std::printf("cxx_host_hp: order %d  meshes %d  levels %d  ndofs %d  coarsest %d  aux %d  coarse %s  iterations %d  converged %d  "
                "|b - A x| / |b| %.3e  sum(x) %.12e\n",
                order, nmesh, nlev, n, nd_fespaces.GetFESpaceAtLevel(0).GetTrueVSize(), (int)aux, coarse.c_str(),
                ksp.GetKrylovSolver().GetNumIterations(), (int)ksp.GetKrylovSolver().GetConverged(),
                linalg::Norml2(ctx, res) / linalg::Norml2(ctx, rhs), linalg::Dot(ctx, x, ones));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "palace_amd: %s\n", e.what());
    return 1;
  }
  return 0;
}
