// Compiles and runs the Palace-side shim printed in INTEGRATION.md section 1 (tests/test_integration_shim.py extracts that
// code block into integration_shim.hpp next to this file's build) against fake_mfem.hpp.
//   integration_shim_check                 no device: the calls that touch none (create, Size, DestroyAssemblyData, destroy)
//   integration_shim_check problem.bin     device: y = A x, y += A x, y = A^T x, the diagonal and the multiplicity-scaled
//                                          forms for the operator described by the blobs; results written to problem.bin.out
#include "fake_mfem.hpp"

#include <cstring>
#include <fstream>

#include "integration_shim.hpp"

static std::vector<std::vector<char>> read_blobs(const char *path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) MFEM_ABORT("cannot open the problem file");
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
  using palace::Vector;
  if (argc < 2) {
    palace::amd::Operator op(10, 12);
    if (op.Size() != 0 || op.Height() != 10 || op.Width() != 12) return 1;
    op.DestroyAssemblyData();
    op.SetDofMultiplicity(Vector());
    std::printf("shim ok (no device): %s\n", pa_version());
    return 0;
  }
  auto blobs = read_blobs(argv[1]);
  auto i32 = [&](size_t i) { return reinterpret_cast<const int32_t *>(blobs[i].data()); };
  auto f64 = [&](size_t i) { return reinterpret_cast<const double *>(blobs[i].data()); };
  // blob 0: {ne, mesh_order, q1d, num_nodes, P, lsize, fe_type, order, qfunction, trial_ops, test_ops}
  const int32_t *h = i32(0);
  pa_mesh_desc m{h[0], h[1], h[2], h[3], i32(1), f64(2), i32(3), f64(4), f64(5), f64(6)};
  pa_geom *geom = nullptr;
  if (pa_geom_create(&m, nullptr, &geom)) MFEM_ABORT(pa_last_error());
  pa_restriction_desc r{h[0], h[4], h[5], i32(7), reinterpret_cast<const uint8_t *>(blobs[8].data()), nullptr};
  pa_basis_desc b{h[6], h[7], h[2], f64(9), f64(10), f64(11), i32(12), nullptr, nullptr};
  std::vector<CeedIntScalar> ctx(blobs[13].size() / 8);
  std::memcpy(ctx.data(), blobs[13].data(), blobs[13].size());
  const int n = h[5];
  palace::amd::Operator op(n, n);
  op.AddSubOperator(geom, r, b, (pa_qfunction)h[8], ctx, (unsigned)h[9], (unsigned)h[10]);
  op.Finalize();
  if (op.Size() != 1) return 1;
  Vector x(n), y(n), y2(n), yt(n), diag(n), ym(n), yma(n), ymt(n), d(n);
  std::memcpy(x.HostReadWrite(), f64(14), (size_t)n * 8);
  std::memcpy(d.HostReadWrite(), f64(15), (size_t)n * 8);
  op.Mult(x, y);
  y2 = y;
  op.AddMult(x, y2);  // = 2 A x
  op.MultTranspose(x, yt);
  op.AssembleDiagonal(diag);
  op.DestroyAssemblyData();
  op.SetDofMultiplicity(Vector(d));
  op.Mult(x, ym);             // d .* (A x)
  yma = x;
  op.AddMult(x, yma);         // x + d .* (A x)
  op.MultTranspose(x, ymt);   // A^T (d .* x)
  std::ofstream o(std::string(argv[1]) + ".out", std::ios::binary);
  for (const Vector *v : {&y, &y2, &yt, &diag, &ym, &yma, &ymt}) o.write(reinterpret_cast<const char *>(v->HostRead()), (size_t)n * 8);
  pa_geom_destroy(geom);
  std::printf("shim ok (device): %d dofs\n", n);
  return 0;
}
