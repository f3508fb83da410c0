// C++ host example: one bilinear form with domain and boundary integrators, the way SpaceOperator composes the system
// matrix of a driven simulation with impedance / London-depth type boundaries (models/spaceoperator.cpp:270-303):
//   a.AddDomainIntegrator<CurlCurlMassIntegrator>(mu^-1, eps)
//   a.AddBoundaryIntegrator<VectorFEMassIntegrator>(sigma)   (f_apply_hcurl_32 on the boundary triangles)
//   a.AddBoundaryIntegrator<CurlCurlIntegrator>(lambda)      (f_apply_l2_1 on the scalar surface curl)
// on a tetrahedral Nedelec space given by dense tables (see dump_boundary_problem.py); applies it to a vector, assembles the
// diagonal and writes both.
//   ./boundary_form problem.bin out.bin
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "fem.hpp"

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
    const int32_t *h = i32(0);
    const int ne = h[0], npe = h[1], nq = h[2], nn = h[3], p = h[4], nd_size = h[5], nd_P = h[6], nd_diag = h[7], bne = h[8],
              bnpe = h[9], bnq = h[10], bnn = h[11], b_P = h[12];
    hipStream_t stream;
    if (hipStreamCreate(&stream) != hipSuccess) throw pa::Error("no HIP device");
    Context ctx;
    ctx.stream = stream;

    pa_mesh_dense_desc md{ne, npe, nq, nn, i32(1), f64(2), i32(3), f64(4), f64(5), 3, 0};
    Mesh mesh(ctx, md);
    FiniteElementSpace nd(ctx, mesh, PA_FE_HCURL, p, nd_P, nd_size, i32(6), nd_diag ? u8(7) : nullptr,
                          nd_diag ? nullptr : reinterpret_cast<const int8_t *>(blobs[8].data()), f64(9), f64(10));
    pa_mesh_dense_desc bd{bne, bnpe, bnq, bnn, i32(11), f64(12), i32(13), f64(14), f64(15), 2, 3};
    Mesh bdr_mesh(ctx, bd);
    FiniteElementSpace nd_bdr(ctx, bdr_mesh, PA_FE_HCURL, p, b_P, nd_size, i32(16), u8(17), nullptr, f64(18), f64(19));

    const std::vector<int> attr_mat{0, 1};
    MaterialPropertyCoefficient muinv(attr_mat, 3, std::vector<double>(f64(20), f64(20) + 18)),
        eps(attr_mat, 3, std::vector<double>(f64(21), f64(21) + 18)), sigma(attr_mat, 3, std::vector<double>(f64(22), f64(22) + 18)),
        lambda(attr_mat, 1, std::vector<double>(f64(23), f64(23) + 2));

    BilinearForm a(nd);
    a.AddDomainIntegrator<CurlCurlMassIntegrator>(muinv, eps);
    a.AddBoundaryIntegrator<VectorFEMassIntegrator>(nd_bdr, sigma);
    a.AddBoundaryIntegrator<CurlCurlIntegrator>(nd_bdr, lambda);
    auto A = a.PartialAssemble();

    Vector x(nd_size), y(nd_size), d(nd_size);
    hipMemcpy(x.Data(), f64(24), sizeof(double) * nd_size, hipMemcpyHostToDevice);
    A->Mult(x, y);
    A->AssembleDiagonal(d);
    hipStreamSynchronize(stream);
    std::vector<double> out((size_t)2 * nd_size);
    hipMemcpy(out.data(), y.Data(), sizeof(double) * nd_size, hipMemcpyDeviceToHost);
    hipMemcpy(out.data() + nd_size, d.Data(), sizeof(double) * nd_size, hipMemcpyDeviceToHost);
    std::ofstream(argv[2], std::ios::binary).write(reinterpret_cast<const char *>(out.data()), sizeof(double) * out.size());
    std::printf("tets %d boundary triangles %d dofs %d symmetric %d\nOK\n", ne, bne, nd_size, (int)A->IsSymmetric());
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
