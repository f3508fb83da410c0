// C++ host example: the classes of palace_amd/csrc/linalg.hpp used the way Palace's drivers use
// palace::ceed::Operator / ParOperator / CgSolver / ChebyshevSmoother, on descriptor arrays that come
// from a file instead of MFEM.  Solves (K + eps M) x = b on a small PEC cavity and prints the
// iteration count, the final residual and a checksum of x.
//
//   hipcc --offload-arch=gfx950 -std=c++17 -I<repo>/palace_amd/csrc solve.cpp -L<repo>/palace_amd/lib
//         -lpalace_amd -Wl,-rpath,<repo>/palace_amd/lib -o solve
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "linalg.hpp"

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

static void check(int rc) {
  if (rc) {
    std::fprintf(stderr, "palace_amd: %s\n", pa_last_error());  // ceed.hpp:13-33 convention -> MFEM_ABORT in Palace
    std::exit(1);
  }
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  auto blobs = read_blobs(argv[1]);
  const auto *hdr = reinterpret_cast<const int32_t *>(blobs[0].data());
  const int ne = hdr[0], P = hdr[1], ndofs = hdr[2], p = hdr[3], q1d = hdr[4], nn = hdr[5];
  auto as = [&](int i) { return blobs[(size_t)i].data(); };

  // what InitRestriction / InitBasis / AssembleGeometryData fill for libCEED today
  pa_restriction_desc r{ne, P, ndofs, (const int32_t *)as(1), (const uint8_t *)as(2), nullptr};
  pa_basis_desc b{PA_FE_HCURL, p, q1d, (const double *)as(4), (const double *)as(5), (const double *)as(6),
                  (const int32_t *)as(3), nullptr, nullptr};
  pa_mesh_desc m{ne, 2, q1d, nn, (const int32_t *)as(7), (const double *)as(8), (const int32_t *)as(9),
                 (const double *)as(10), (const double *)as(11), (const double *)as(12)};
  pa_geom *geom = nullptr;
  check(pa_geom_create(&m, nullptr, &geom));
  pa_op *op = nullptr;
  check(pa_op_create(ndofs, ndofs, &op));
  check(pa_op_add_sub(op, geom, &r, &b, PA_QF_HDIVMASS_33, as(13), blobs[13].size(), PA_EVAL_CURL | PA_EVAL_INTERP,
                      PA_EVAL_CURL | PA_EVAL_INTERP));
  check(pa_op_finalize(op));

  Context ctx;  // default stream, single process
  {
    ceed::Operator local(ctx, op, /*own=*/true);
    const int n_ess = (int)(blobs[14].size() / 4);
    ParOperator A(ctx, local, ndofs, (const int32_t *)as(14), n_ess, ParOperator::DiagonalPolicy::DIAG_ONE);

    ChebyshevSmoother smoother(ctx, /*smooth_it=*/1, /*order=*/4);
    smoother.SetOperator(A);
    CgSolver pcg(ctx);
    pcg.SetOperator(A);
    pcg.SetPreconditioner(smoother);
    pcg.SetTol(1e-10);
    pcg.SetMaxIter(500);

    Vector ones(ndofs), rhs(ndofs), x(ndofs);
    linalg::Fill(ctx, ones, 1.0);
    A.Mult(ones, rhs);
    linalg::SetSubVector(ctx, rhs, A.GetEssentialTrueDofs(), A.NumEssentialTrueDofs(), 0.0);
    pcg.Mult(rhs, x);

    Vector res(ndofs);
    A.Mult(x, res);
    linalg::AXPBY(ctx, 1.0, rhs, -1.0, res);
    std::printf("cxx_host: ndofs %d  iterations %d  converged %d  |b - A x| / |b| %.3e  sum(x) %.12e\n", ndofs,
                pcg.GetNumIterations(), (int)pcg.GetConverged(), linalg::Norml2(ctx, res) / linalg::Norml2(ctx, rhs),
                linalg::Dot(ctx, x, ones));
  }
  pa_geom_destroy(geom);
  return 0;
}
