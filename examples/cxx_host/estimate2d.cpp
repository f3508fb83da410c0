// C++ host example: the plane branch of the flux error estimators (linalg/errorestimator.cpp:343-349 and :446-472).  From
// the arrays of dump_estimator_problem_2d.py it builds a dense 2-D Mesh, the Nedelec, Raviart-Thomas, H1 and discontinuous
// scalar FiniteElementSpaces and runs
//   GradFluxErrorEstimator(eps [2 x 2], nd, rt)      -- f_apply_hcurlhdiv_22 flux, f_apply_hdiv_22 mass, f_apply_hcurlhdiv_error_22
//   CurlFluxErrorEstimator(muinv [1 x 1], l2, h1)    -- MassIntegrator flux and mass (scalar_flux), f_apply_l2h1_error
// into ErrorIndicators; writes the two indicator vectors to a file.
//   ./estimate2d problem.bin out.bin
#include <cstdio>
#include <cstdlib>
#include <fstream>
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
    auto u8 = [&](size_t i) { return reinterpret_cast<const uint8_t *>(blobs[i].data()); };
    const int ne = i32(0)[0], npe = i32(0)[1], nq = i32(0)[2], nn = i32(0)[3], p = i32(0)[4], nd_size = i32(0)[5],
              nd_P = i32(0)[6], h1_size = i32(0)[7], h1_P = i32(0)[8], l2_size = i32(0)[9];
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) throw pa::Error("no HIP device");
    Context ctx;
    ctx.stream = stream;

    pa_mesh_dense_desc md{ne, npe, nq, nn, i32(1), f64(2), i32(3), f64(4), f64(5), 2, 0};
    Mesh mesh(ctx, md);
    FiniteElementSpace nd(ctx, mesh, PA_FE_HCURL, p, nd_P, nd_size, i32(6), u8(7), nullptr, f64(8), f64(9));
    FiniteElementSpace rt(ctx, mesh, PA_FE_HDIV, p, nd_P, nd_size, i32(6), u8(7), nullptr, f64(10), nullptr);
    FiniteElementSpace h1(ctx, mesh, PA_FE_H1, p, h1_P, h1_size, i32(11), nullptr, nullptr, f64(12), f64(13));
    FiniteElementSpace l2(ctx, mesh, PA_FE_H1, p, h1_P, l2_size, i32(14), nullptr, nullptr, f64(12), f64(13));
    MaterialTensors eps{{0, 1}, std::vector<double>(f64(15), f64(15) + 8), 2};
    MaterialTensors muinv{{0, 1}, std::vector<double>(f64(16), f64(16) + 2), 1};

    Vector E(nd_size), B(l2_size);
    hipMemcpy(E.Data(), f64(17), sizeof(double) * nd_size, hipMemcpyHostToDevice);
    hipMemcpy(B.Data(), f64(18), sizeof(double) * l2_size, hipMemcpyHostToDevice);

    const double tol = 1e-12;
    GradFluxErrorEstimator grad(eps, nd, rt, tol, 2000, 0);
    CurlFluxErrorEstimator curl(muinv, l2, h1, tol, 2000, 0);
    ErrorIndicator ig(ctx), ic(ctx);
    const double Et = 0.37;
    grad.AddErrorIndicator(E, Et, ig);
    curl.AddErrorIndicator(B, Et, ic);
    std::printf("elements %d nd %d h1 %d l2 %d\n", ne, nd_size, h1_size, l2_size);
    std::printf("grad: norm %.15e pcg_its %d\n", ig.Norml2(), grad.GetProjector().NumIterations());
    std::printf("curl: norm %.15e pcg_its %d\n", ic.Norml2(), curl.GetProjector().NumIterations());
    std::vector<double> out((size_t)2 * ne);
    hipStreamSynchronize(stream);
    hipMemcpy(out.data(), ig.Local().Data(), sizeof(double) * ne, hipMemcpyDeviceToHost);
    hipMemcpy(out.data() + ne, ic.Local().Data(), sizeof(double) * ne, hipMemcpyDeviceToHost);
    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char *>(out.data()), sizeof(double) * out.size());
    std::printf("OK\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
