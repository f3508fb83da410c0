// Host-only pieces of the C++ front end (palace_amd/csrc/fem.hpp, ksp.hpp) printed for tests/test_fem_host.py, which
// compares them with the numpy restatements the rest of the test-suite is built on: 1-D point sets and Lagrange tables,
// MaterialPropertyCoefficient bookkeeping (materialoperator.cpp:586-868) and the QFunction coefficient contexts
// (coefficient.cpp:51-131), the p-coarsening sequences (multigrid.hpp:44-69).  No device is touched.
#include <cmath>
#include <cstdio>
#include <cstring>

#include <memory>
#include "amg.hpp"
#include "amg_dist.hpp"
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

namespace {
// 5-point operator -ex u_xx - ey u_yy on an n x n grid, homogeneous Dirichlet data
palace::amg::HostCsr grid_laplacian(int n, double ex, double ey) {
  palace::amg::HostCsr A;
  A.nrows = A.ncols = n * n;
  A.rowptr.push_back(0);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) {
      auto put = [&](int ii, int jj, double v) {
        if (ii >= 0 && ii < n && jj >= 0 && jj < n) A.col.push_back(jj * n + ii), A.val.push_back(v);
      };
      put(i, j - 1, -ey), put(i - 1, j, -ex), put(i, j, 2 * ex + 2 * ey), put(i + 1, j, -ex), put(i, j + 1, -ey);
      A.rowptr.push_back((int)A.col.size());
    }
  return A;
}
std::vector<double> dense_of(const palace::amg::HostCsr &A) {
  std::vector<double> d((size_t)A.nrows * A.ncols, 0.0);
  for (int r = 0; r < A.nrows; r++)
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++) d[(size_t)r * A.ncols + A.col[a]] = A.val[a];
  return d;
}
// one V-cycle of the hierarchy: nu weighted-Jacobi sweeps before and after, dense elimination on the last level
void vcycle(const palace::amg::Hierarchy &h, size_t l, const std::vector<double> &b, std::vector<double> &x) {
  const auto &A = h.A[l];
  const int n = A.nrows;
  if (l + 1 == h.A.size()) {
    std::vector<double> M = dense_of(A), r = b;
    for (int k = 0; k < n; k++)
      for (int i = k + 1; i < n; i++) {
        const double f = M[(size_t)i * n + k] / M[(size_t)k * n + k];
        for (int j = k; j < n; j++) M[(size_t)i * n + j] -= f * M[(size_t)k * n + j];
        r[i] -= f * r[k];
      }
    x.assign(n, 0.0);
    for (int i = n - 1; i >= 0; i--) {
      double s = r[i];
      for (int j = i + 1; j < n; j++) s -= M[(size_t)i * n + j] * x[j];
      x[i] = s / M[(size_t)i * n + i];
    }
    return;
  }
  std::vector<double> d(n), t;
  for (int r = 0; r < n; r++)
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++)
      if (A.col[a] == r) d[r] = A.val[a];
  auto smooth = [&]() {
    for (int it = 0; it < 2; it++) {
      palace::amg::Mult(A, x, t);
      for (int i = 0; i < n; i++) x[i] += (2.0 / 3.0) * (b[i] - t[i]) / d[i];
    }
  };
  smooth();
  palace::amg::Mult(A, x, t);
  std::vector<double> res(n), rc, ec, corr;
  for (int i = 0; i < n; i++) res[i] = b[i] - t[i];
  palace::amg::Mult(palace::amg::Transpose(h.P[l]), res, rc);
  ec.assign(rc.size(), 0.0);
  vcycle(h, l + 1, rc, ec);
  palace::amg::Mult(h.P[l], ec, corr);
  for (int i = 0; i < n; i++) x[i] += corr[i];
  smooth();
}
}  // namespace

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
  {  // smoothed-aggregation set-up (amg.hpp): small problem dumped densely, larger ones through their V-cycle convergence
    using namespace palace::amg;
    const HostCsr A = grid_laplacian(10, 1.0, 1.0);
    int na = 0;
    const auto agg = Aggregate(A, 0.08, na);
    const HostCsr T = TentativeProlongator(agg, na);
    const Hierarchy h = Setup(A, 2, 10);
    std::printf("amg_small %d %d %d\n", na, (int)h.A.size(), h.A.back().nrows);
    dump("amg_T", dense_of(T));
    dump("amg_P", dense_of(h.P.at(0)));
    dump("amg_A1", dense_of(h.A.at(1)));
    for (int cas = 0; cas < 2; cas++) {
      const HostCsr B = grid_laplacian(48, 1.0, cas == 0 ? 1.0 : 0.01);  // isotropic, then strongly anisotropic
      const Hierarchy hb = Setup(B, 10, 60);
      std::printf("amg_levels%d", cas);
      for (const auto &L : hb.A) std::printf(" %d", L.nrows);
      std::printf("\n");
      std::vector<double> b(B.nrows), x(B.nrows, 0.0), t, fac;
      for (int i = 0; i < B.nrows; i++) b[i] = std::sin(0.37 * i) + 0.5;
      double prev = 0.0;
      for (double v : b) prev += v * v;
      prev = std::sqrt(prev);
      for (int it = 0; it < 12; it++) {
        std::vector<double> r(B.nrows), e(B.nrows, 0.0);
        Mult(B, x, t);
        for (int i = 0; i < B.nrows; i++) r[i] = b[i] - t[i];
        vcycle(hb, 0, r, e);
        for (int i = 0; i < B.nrows; i++) x[i] += e[i];
        Mult(B, x, t);
        double nr = 0.0;
        for (int i = 0; i < B.nrows; i++) nr += (b[i] - t[i]) * (b[i] - t[i]);
        nr = std::sqrt(nr);
        fac.push_back(nr / prev);
        prev = nr;
      }
      dump(cas == 0 ? "amg_factors_iso" : "amg_factors_aniso", fac);
    }
  }
  {  // the hierarchy of the distributed solve (amg.hpp: SetupBlocks; amg_dist.hpp: DistSpace): aggregates confined to the ranks' row
     // blocks and numbered block by block; every rank's rows of A_l, R_l, P_l in its local numbering [own | ghosts] times the
     // local copy of a global vector = the rows of the global product; V-cycle convergence beside the unconfined hierarchy's
    using namespace palace::amg;
    const HostCsr A = grid_laplacian(40, 1.0, 0.6);
    const std::vector<int> off = {0, 530, 1111, 1600};
    const int size = 3;
    int na = 0;
    std::vector<int> aoff;
    const std::vector<int> agg = AggregateBlocks(A, 0.08, off, na, aoff);
    int confined = (int)aoff.size() == size + 1 && aoff.back() == na;
    for (int b = 0; b < size && confined; b++)
      for (int i = off[b]; i < off[b + 1]; i++) confined = confined && (agg[i] < 0 || (agg[i] >= aoff[b] && agg[i] < aoff[b + 1]));
    std::vector<std::vector<int>> loff;
    const Hierarchy h = SetupBlocks(A, off, loff, 10, 60);
    const Hierarchy hg = Setup(A, 10, 60);
    std::printf("amg_blocks %d %d %d %d\n", confined, (int)h.A.size(), h.A.back().nrows, (int)hg.A.size());
    {  // thin partitions (round-5 advisor finding): strong ties along y only, row blocks of one and of two grid lines -- every strong
       // neighbour of a row of a one-line block belongs to another block.  No row may drop out of the coarse space for that.
      const HostCsr T = grid_laplacian(24, 0.05, 1.0);
      int dropped[2] = {0, 0}, nagg[2] = {0, 0};
      for (int lines = 1; lines <= 2; lines++) {
        std::vector<int> toff;
        for (int r = 0; r <= 24 * 24; r += 24 * lines) toff.push_back(r);
        std::vector<int> taoff;
        const std::vector<int> tagg = AggregateBlocks(T, 0.08, toff, nagg[lines - 1], taoff);
        for (int a : tagg) dropped[lines - 1] += a < 0;
      }
      std::printf("amg_thin_blocks %d %d %d %d\n", dropped[0], nagg[0], dropped[1], nagg[1]);
    }
    double worst = 0.0;
    long long ghosts = 0, plan_mismatch = 0, plan_entries = 0;
    const size_t nl = h.A.size();
    // the exchange plans of level 0 as every rank derives them on its own: what rank r sends to s is, entry by entry, what s
    // expects from r (the transport matches the two sides of a piece by position)
    {
      std::vector<std::unique_ptr<palace::DistSpace>> sp(size);
      std::vector<std::vector<int>> nbr(size), so(size), ro(size);
      std::vector<std::vector<int32_t>> si(size), ri(size);
      const HostCsr R0 = Transpose(h.P[0]);
      for (int r = 0; r < size; r++) {
        sp[r] = std::make_unique<palace::DistSpace>(r, size, loff[0]);
        sp[r]->Need(h.A[0], loff[0]);
        sp[r]->Need(R0, loff[1]);
        sp[r]->Finalize(nullptr);
        sp[r]->Plan(nbr[r], so[r], si[r], ro[r], ri[r]);
      }
      for (int r = 0; r < size; r++)
        for (size_t k = 0; k < nbr[r].size(); k++) {
          const int s = nbr[r][k];
          size_t j = 0;
          while (j < nbr[s].size() && nbr[s][j] != r) j++;
          if (j == nbr[s].size() || so[r][k + 1] - so[r][k] != ro[s][j + 1] - ro[s][j]) {
            plan_mismatch++;
            continue;
          }
          for (int i = 0; i < so[r][k + 1] - so[r][k]; i++) {
            const int sent = sp[r]->Offset() + si[r][so[r][k] + i];                              // global number of what r sends
            const int slot = ri[s][ro[s][j] + i] - sp[s]->NumOwned();                            // ghost slot s fills with it
            plan_mismatch += (slot < 0 || slot >= sp[s]->NumGhosts() || sp[s]->Ghosts()[slot] != sent) ? 1 : 0;
            plan_entries++;
          }
        }
    }
    std::vector<HostCsr> R(nl - 1);
    for (size_t l = 0; l + 1 < nl; l++) R[l] = Transpose(h.P[l]);
    for (int rank = 0; rank < size; rank++) {
      std::vector<std::unique_ptr<palace::DistSpace>> sp(nl);
      for (size_t l = 0; l < nl; l++) {
        sp[l] = std::make_unique<palace::DistSpace>(rank, size, loff[l]);
        sp[l]->Need(h.A[l], loff[l]);
        if (l + 1 < nl) sp[l]->Need(R[l], loff[l + 1]);
        if (l > 0) sp[l]->Need(h.P[l - 1], loff[l - 1]);
        sp[l]->Finalize(nullptr);
        ghosts += sp[l]->NumGhosts();
      }
      auto check = [&](const HostCsr &M, const std::vector<int> &row_off, const palace::DistSpace &cols) {
        const HostCsr L = cols.Localize(M, row_off);
        std::vector<double> xg((size_t)M.ncols), yg, xl((size_t)cols.NumLocal()), yl;
        for (int i = 0; i < M.ncols; i++) xg[i] = std::sin(0.7 * i + 0.1 * rank) + 0.3;
        for (int i = 0; i < cols.NumOwned(); i++) xl[i] = xg[cols.Offset() + i];
        for (int k = 0; k < cols.NumGhosts(); k++) xl[cols.NumOwned() + k] = xg[cols.Ghosts()[k]];
        Mult(M, xg, yg), Mult(L, xl, yl);
        for (int r = 0; r < L.nrows; r++) worst = std::max(worst, std::abs(yl[r] - yg[row_off[rank] + r]));
      };
      for (size_t l = 0; l < nl; l++) {
        check(h.A[l], loff[l], *sp[l]);
        if (l + 1 < nl) check(R[l], loff[l + 1], *sp[l]), check(h.P[l], loff[l], *sp[l + 1]);
      }
    }
    dump("amg_blocks_products", std::vector<double>{worst, (double)ghosts, (double)plan_mismatch, (double)plan_entries});
    for (int cas = 0; cas < 2; cas++) {
      const Hierarchy &hh = cas ? hg : h;
      std::vector<double> b(A.nrows), x(A.nrows, 0.0), t, fac;
      for (int i = 0; i < A.nrows; i++) b[i] = std::sin(0.37 * i) + 0.5;
      double prev = 0.0;
      for (double v : b) prev += v * v;
      prev = std::sqrt(prev);
      for (int it = 0; it < 10; it++) {
        std::vector<double> r(A.nrows), e(A.nrows, 0.0);
        Mult(A, x, t);
        for (int i = 0; i < A.nrows; i++) r[i] = b[i] - t[i];
        vcycle(hh, 0, r, e);
        for (int i = 0; i < A.nrows; i++) x[i] += e[i];
        Mult(A, x, t);
        double nr = 0.0;
        for (int i = 0; i < A.nrows; i++) nr += (b[i] - t[i]) * (b[i] - t[i]);
        nr = std::sqrt(nr);
        fac.push_back(nr / prev);
        prev = nr;
      }
      dump(cas ? "amg_factors_global" : "amg_factors_blocks", fac);
    }
  }
  {  // the set-up is row-parallel (PALACE_AMD_SETUP_THREADS): a problem with many row blocks, every matrix of its hierarchy
     // summed with position-dependent weights -- the test runs this program with one and with several threads and compares
    using namespace palace::amg;
    const Hierarchy h = Setup(grid_laplacian(160, 1.0, 0.3), 10, 60);
    double cs = 0.0;
    long long nnz = 0;
    for (const std::vector<HostCsr> *v : {&h.A, &h.P})
      for (const HostCsr &M : *v) {
        nnz += M.nnz();
        for (int r = 0; r < M.nrows; r++)
          for (int a = M.rowptr[r]; a < M.rowptr[r + 1]; a++) cs += M.val[a] * (1.0 + 1e-3 * ((r + 3 * M.col[a]) % 101));
      }
    std::printf("amg_threads_checksum %.17g %lld %d\n", cs, nnz, (int)h.A.size());
  }
  return 0;
}
