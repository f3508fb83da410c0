// Host-only pieces of the C++ front end (palace_amd/csrc/fem.hpp, ksp.hpp) printed for tests/test_fem_host.py, which
// compares them with the numpy restatements the rest of the test-suite is built on: 1-D point sets and Lagrange tables,
// MaterialPropertyCoefficient bookkeeping (materialoperator.cpp:586-868) and the QFunction coefficient contexts
// (coefficient.cpp:51-131), the p-coarsening sequences (multigrid.hpp:44-69).  No device is touched.
#include <cstdio>
#include <cstring>

#include "errorestimator.hpp"
#include "ksp.hpp"

using namespace palace;

static void dump(const char *name, const std::vector<double> &v) {
  std::printf("%s", name);
  for (double x : v) {
    uint64_t w;
    std::memcpy(&w, &x, 8);
    std::printf(" %016llx", (unsigned long long)w);
  }
  std::printf("\n");
}

int main() {
  for (int n = 1; n <= 6; n++) {
    std::vector<double> x, w;
    fem::GaussLegendre(n, x, w);
    dump(("gl_x" + std::to_string(n)).c_str(), x);
    dump(("gl_w" + std::to_string(n)).c_str(), w);
    if (n >= 2) dump(("gll" + std::to_string(n)).c_str(), fem::GaussLobatto(n));
  }
  {
    std::vector<double> x, w, B, G;
    fem::GaussLegendre(4, x, w);
    fem::LagrangeEval(fem::GaussLobatto(4), x, B, G);
    dump("Bc3", B), dump("Gc3", G);
  }
  // identity (no coefficient), scaled
  dump("ctx_identity", ceed::PopulateCoefficientContext(3, nullptr, false, 1.5));
  {
    MaterialPropertyCoefficient Q(3);
    Q.AddMaterialProperty(1, 2.08);
    Q.AddMaterialProperty(std::vector<int>{3}, 2.08);  // equal material is reused
    dump("ctx_scalar", ceed::PopulateCoefficientContext(3, &Q));
    const double M[9] = {2.0, 0.1, 0.2, 0.3, 3.0, 0.4, 0.5, 0.6, 4.0};  // column-major
    Q.AddMaterialProperty(std::vector<int>{2}, M, 3, 0.5);  // scalars become diagonals
    dump("ctx_mixed", ceed::PopulateCoefficientContext(3, &Q));
    dump("ctx_mixed_t", ceed::PopulateCoefficientContext(3, &Q, true, 2.0));
    Q.AddMaterialProperty(std::vector<int>{1, 3}, 1.0, -1.0);  // update in place: 2.08 - 1
    dump("ctx_updated", ceed::PopulateCoefficientContext(3, &Q));
    Q.RestrictCoefficient({2, 3});
    dump("ctx_restricted", ceed::PopulateCoefficientContext(3, &Q));
    MaterialPropertyCoefficient Qm(3);
    Qm.AddMaterialProperty(std::vector<int>{1, 2, 3}, 0.7);
    dump("ctx_pair", ceed::PopulateCoefficientContext(1, &Qm, 3, &Q));
    Q.NormalProjectedCoefficient({0.0, 0.6, 0.8});
    dump("ctx_normal", ceed::PopulateCoefficientContext(1, &Q));
  }
  for (int p = 1; p <= 6; p++) {
    std::printf("orders_log%d", p);
    for (int q : GetPolynomialOrders(p, MultigridCoarsening::LOGARITHMIC)) std::printf(" %d", q);
    std::printf("\norders_lin%d", p);
    for (int q : GetPolynomialOrders(p, MultigridCoarsening::LINEAR)) std::printf(" %d", q);
    std::printf("\n");
  }
  fem::DefaultIntegrationOrder::p_trial = 3;
  std::printf("q1d %d\n", fem::DefaultIntegrationOrder::GetQ1d(5));
  {  // linalg::MatrixSqrt / MatrixPow on the material tensors of the estimators (densematrix.cpp:222-252)
    const double S[9] = {2.0, 0.3, 0.0, 0.3, 1.5, 0.1, 0.0, 0.1, 1.2};
    const auto r = linalg::MatrixSqrt(S), ir = linalg::MatrixPow(S, -0.5);
    dump("mat_sqrt", std::vector<double>(r.begin(), r.end()));
    dump("mat_invsqrt", std::vector<double>(ir.begin(), ir.end()));
    const double D[9] = {4.0, 0, 0, 0, 9.0, 0, 0, 0, 0.25};  // already diagonal, and a repeated eigenvalue
    const auto rd = linalg::MatrixSqrt(D);
    dump("mat_sqrt_diag", std::vector<double>(rd.begin(), rd.end()));
    const double R[9] = {3.0, 1.0, 0, 1.0, 3.0, 0, 0, 0, 2.0};  // eigenvalues 2, 2, 4
    const auto rr = linalg::MatrixPow(R, 2.0);
    dump("mat_square", std::vector<double>(rr.begin(), rr.end()));
    // MaterialTensors of a plane problem: 2 x 2 permittivities and the 1 x 1 curl-curl inverse permeability
    // (errorestimator.cpp:331-336, :452-459); the matrix functions see them bordered with an identity block
    const MaterialTensors eps2{{0, 1}, {2.0, 0.3, 0.3, 1.5, 3.1, 0.0, 0.0, 3.1}, 2};
    const auto s2 = eps2.Map([](const double *m) { return linalg::MatrixSqrt(m); });
    const auto i2 = eps2.Map([](const double *m) { return linalg::MatrixPow(m, -0.5); });
    dump("mat2_sqrt", s2.mat);
    dump("mat2_invsqrt", i2.mat);
    const MaterialTensors mu1{{0, 1}, {0.8, 1.4}, 1};
    dump("mat1_sqrt", mu1.Map([](const double *m) { return linalg::MatrixSqrt(m); }).mat);
    std::printf("mat_dims %d %d\n", s2.dim, (int)s2.mat.size());
  }
  return 0;
}
