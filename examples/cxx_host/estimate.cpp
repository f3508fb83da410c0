// C++ host example: the flux error estimators of a Palace post-processing step without Python and without MFEM.  From the
// arrays MFEM would provide for a tetrahedral mesh (see dump_estimator_problem.py) it builds a dense Mesh, the Nedelec and
// Raviart-Thomas FiniteElementSpaces, MaterialTensors for eps and mu^-1, and runs
//   GradFluxErrorEstimator::AddErrorIndicator(E), CurlFluxErrorEstimator::AddErrorIndicator(B),
//   TimeDependentFluxErrorEstimator::AddErrorIndicator(E, B)
// (linalg/errorestimator.cpp:271-541) into ErrorIndicators; prints the indicator norms, PCG iteration counts and writes
// the three indicator vectors to a file.  With the H1 space of the dump it also applies
//   BilinearForm(h1, nd) + MixedVectorGradientIntegrator(eps)   and   BilinearForm(h1, rt) + MixedVectorGradientIntegrator(eps)
// (fem/integ/mixedvecgrad.cpp; the form of models/modeeigensolver.cpp:52) to a potential and writes both results to out.bin.grad.
//   ./estimate problem.bin out.bin
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "errorestimator.hpp"

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
  if (argc < 3) return 2;
  try {
    auto blobs = read_blobs(argv[1]);
    auto i32 = [&](size_t i) { return reinterpret_cast<const int32_t *>(blobs[i].data()); };
    auto f64 = [&](size_t i) { return reinterpret_cast<const double *>(blobs[i].data()); };
    const int ne = i32(0)[0], npe = i32(0)[1], nq = i32(0)[2], nn = i32(0)[3], p = i32(0)[4], nd_size = i32(0)[5],
              nd_P = i32(0)[6], rt_size = i32(0)[7], rt_P = i32(0)[8], nd_diag = i32(0)[9];
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) throw pa::Error("no HIP device");
    Context ctx;
    ctx.stream = stream;

    pa_mesh_dense_desc md{ne, npe, nq, nn, i32(1), f64(2), i32(3), f64(4), f64(5), 3, 0};
    Mesh mesh(ctx, md);
    FiniteElementSpace nd(ctx, mesh, PA_FE_HCURL, p, nd_P, nd_size, i32(6),
                          nd_diag ? reinterpret_cast<const uint8_t *>(blobs[7].data()) : nullptr,
                          nd_diag ? nullptr : reinterpret_cast<const int8_t *>(blobs[8].data()), f64(9), f64(10));
    FiniteElementSpace rt(ctx, mesh, PA_FE_HDIV, p, rt_P, rt_size, i32(11), reinterpret_cast<const uint8_t *>(blobs[12].data()),
                          nullptr, f64(13), nullptr);
    MaterialTensors eps{{0, 1}, std::vector<double>(f64(14), f64(14) + 18)};
    MaterialTensors muinv{{0, 1}, std::vector<double>(f64(15), f64(15) + 18)};

    Vector E(nd_size), B(rt_size);
    hipMemcpy(E.Data(), f64(16), sizeof(double) * nd_size, hipMemcpyHostToDevice);
    hipMemcpy(B.Data(), f64(17), sizeof(double) * rt_size, hipMemcpyHostToDevice);

    const double tol = 1e-12;
    GradFluxErrorEstimator grad(eps, nd, rt, tol, 500, 0);
    CurlFluxErrorEstimator curl(muinv, rt, nd, tol, 500, 0);
    TimeDependentFluxErrorEstimator both(eps, muinv, nd, rt, tol, 500, 0);
    ErrorIndicator ig(ctx), ic(ctx), it(ctx);
    const double Et = 0.37;
    grad.AddErrorIndicator(E, Et, ig);
    curl.AddErrorIndicator(B, Et, ic);
    both.AddErrorIndicator(E, B, Et, it);
    both.AddErrorIndicator(E, B, 0.0, it);  // a second sample: running root mean square (errorindicator.cpp:11-47)
    std::printf("elements %d nd %d rt %d\n", ne, nd_size, rt_size);
    std::printf("grad: norm %.15e pcg_its %d\n", ig.Norml2(), grad.GetProjector().NumIterations());
    std::printf("curl: norm %.15e pcg_its %d\n", ic.Norml2(), curl.GetProjector().NumIterations());
    std::printf("both: norm %.15e samples %d\n", it.Norml2(), it.NumSamples());
    std::vector<double> out((size_t)3 * ne);
    hipStreamSynchronize(stream);
    hipMemcpy(out.data(), ig.Local().Data(), sizeof(double) * ne, hipMemcpyDeviceToHost);
    hipMemcpy(out.data() + ne, ic.Local().Data(), sizeof(double) * ne, hipMemcpyDeviceToHost);
    hipMemcpy(out.data() + 2 * (size_t)ne, it.Local().Data(), sizeof(double) * ne, hipMemcpyDeviceToHost);
    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char *>(out.data()), sizeof(double) * out.size());
    if (blobs.size() >= 23) {
      const int h1_size = i32(18)[0], h1_P = i32(18)[1];
      FiniteElementSpace h1(ctx, mesh, PA_FE_H1, p, h1_P, h1_size, i32(19), nullptr, nullptr, f64(20), f64(21));
      const auto eps_c = eps.Coefficient();
      Vector phi(h1_size), gn(nd_size), gr(rt_size);
      hipMemcpy(phi.Data(), f64(22), sizeof(double) * h1_size, hipMemcpyHostToDevice);
      BilinearForm to_nd(h1, nd), to_rt(h1, rt);
      to_nd.AddDomainIntegrator<MixedVectorGradientIntegrator>(eps_c);
      to_rt.AddDomainIntegrator<MixedVectorGradientIntegrator>(eps_c);
      const auto op_nd = to_nd.PartialAssemble(), op_rt = to_rt.PartialAssemble();
      op_nd->Mult(phi, gn);
      op_rt->Mult(phi, gr);
      hipStreamSynchronize(stream);
      std::vector<double> g((size_t)nd_size + rt_size);
      hipMemcpy(g.data(), gn.Data(), sizeof(double) * nd_size, hipMemcpyDeviceToHost);
      hipMemcpy(g.data() + nd_size, gr.Data(), sizeof(double) * rt_size, hipMemcpyDeviceToHost);
      std::ofstream(std::string(argv[2]) + ".grad", std::ios::binary)
          .write(reinterpret_cast<const char *>(g.data()), sizeof(double) * g.size());
      std::printf("mixed gradient: h1 %d\n", h1_size);
      // GradientIntegrator (fem/integ/grad.cpp): (eps grad phi, v) with v in (H1)^3, both orderings of the vector space
      const GradientIntegrator gi(eps_c);
      std::vector<double> gv((size_t)6 * h1_size);
      for (int by_vdim = 0; by_vdim < 2; by_vdim++) {
        const VectorFiniteElementSpace vh1(h1, 3, by_vdim != 0);
        const auto op_g = gi.PartialAssemble(h1, vh1);
        Vector gy(vh1.GetVSize());
        op_g->Mult(phi, gy);
        hipStreamSynchronize(stream);
        hipMemcpy(gv.data() + (size_t)by_vdim * 3 * h1_size, gy.Data(), sizeof(double) * 3 * h1_size, hipMemcpyDeviceToHost);
      }
      std::ofstream(std::string(argv[2]) + ".vgrad", std::ios::binary)
          .write(reinterpret_cast<const char *>(gv.data()), sizeof(double) * gv.size());
      // MixedVectorWeakDivergenceIntegrator -(eps E, grad v) into H1, and MassIntegrator on (H1)^3 with the 3 x 3 tensor eps
      BilinearForm wd(nd, h1);
      wd.AddDomainIntegrator<MixedVectorWeakDivergenceIntegrator>(eps_c);
      Vector wy(h1_size);
      wd.PartialAssemble()->Mult(E, wy);
      const VectorFiniteElementSpace vh(h1, 3, false);
      const VectorMassIntegrator vm(eps_c);
      Vector vx(3 * h1_size), vy(3 * h1_size);
      hipMemset(vx.Data(), 0, sizeof(double) * 3 * h1_size);
      for (int c = 0; c < 3; c++)  // (phi, 2 phi, 3 phi) as the vector field
        linalg::AXPBY(ctx, c + 1.0, phi, 0.0, *std::make_unique<Vector>(vx.Data() + (size_t)c * h1_size, h1_size));
      vm.PartialAssemble(vh)->Mult(vx, vy);
      hipStreamSynchronize(stream);
      std::vector<double> wv((size_t)4 * h1_size);
      hipMemcpy(wv.data(), wy.Data(), sizeof(double) * h1_size, hipMemcpyDeviceToHost);
      hipMemcpy(wv.data() + h1_size, vy.Data(), sizeof(double) * 3 * h1_size, hipMemcpyDeviceToHost);
      std::ofstream(std::string(argv[2]) + ".wdiv", std::ios::binary)
          .write(reinterpret_cast<const char *>(wv.data()), sizeof(double) * wv.size());
    }
    if (blobs.size() >= 24) {
      // DivDivMassIntegrator (fem/integ/divdivmass.cpp) on the Raviart-Thomas space with its divergence table: (c div u, div v)
      // + (muinv u, v) in one operator, and the same as the sum of DivDivIntegrator and VectorFEMassIntegrator
      FiniteElementSpace rtd(ctx, mesh, PA_FE_HDIV, p, rt_P, rt_size, i32(11), reinterpret_cast<const uint8_t *>(blobs[12].data()),
                             nullptr, f64(13), f64(23));
      const MaterialTensors cdiv{{0, 1}, {1.9, 0.4}, 1};
      const auto q_div = cdiv.Coefficient(), q_mass = muinv.Coefficient();
      BilinearForm pair(rtd), sum(rtd);
      pair.AddDomainIntegrator<DivDivMassIntegrator>(q_div, q_mass);
      sum.AddDomainIntegrator<DivDivIntegrator>(q_div);
      sum.AddDomainIntegrator<VectorFEMassIntegrator>(q_mass);
      const auto op_pair = pair.PartialAssemble(), op_sum = sum.PartialAssemble();
      Vector y1(rt_size), y2(rt_size);
      op_pair->Mult(B, y1);
      op_sum->Mult(B, y2);
      hipStreamSynchronize(stream);
      std::vector<double> yy((size_t)2 * rt_size);
      hipMemcpy(yy.data(), y1.Data(), sizeof(double) * rt_size, hipMemcpyDeviceToHost);
      hipMemcpy(yy.data() + rt_size, y2.Data(), sizeof(double) * rt_size, hipMemcpyDeviceToHost);
      std::ofstream(std::string(argv[2]) + ".divdivmass", std::ios::binary)
          .write(reinterpret_cast<const char *>(yy.data()), sizeof(double) * yy.size());
      std::printf("div-div + mass: rt %d\n", rt_size);
    }
    std::printf("OK\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
