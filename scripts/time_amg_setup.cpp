// Host-only timing of the AMS / AMG set-up (see time_amg_setup.py): the sequence of AmsSolver's constructor
// (amg_solver.hip) -- DropRows, Pi, the four Galerkin products, the two hierarchies -- on matrices read from a file.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "amg.hpp"

using palace::amg::HostCsr;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static HostCsr read_csr(FILE *f) {
  int64_t h[3];
  if (fread(h, 8, 3, f) != 3) std::abort();
  HostCsr A;
  A.nrows = (int)h[0], A.ncols = (int)h[1];
  A.rowptr.resize((size_t)h[0] + 1), A.col.resize((size_t)h[2]), A.val.resize((size_t)h[2]);
  if (fread(A.rowptr.data(), 4, A.rowptr.size(), f) != A.rowptr.size() || fread(A.col.data(), 4, A.col.size(), f) != A.col.size() ||
      fread(A.val.data(), 8, A.val.size(), f) != A.val.size())
    std::abort();
  return A;
}
static HostCsr galerkin(const HostCsr &A, const HostCsr &P) {
  return palace::amg::Multiply(palace::amg::Transpose(P), palace::amg::Multiply(A, P));
}
static double checksum(const HostCsr &A) {
  double s = 0.0;
  for (size_t k = 0; k < A.val.size(); k++) s += A.val[k] * (1.0 + 1e-3 * (A.col[k] % 97));
  return s;
}

int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 1;
  const HostCsr A = read_csr(f), G = read_csr(f);
  int64_t nv;
  if (fread(&nv, 8, 1, f) != 1) return 1;
  std::vector<double> xyz((size_t)3 * nv);
  std::vector<char> ess((size_t)A.nrows);
  if (fread(xyz.data(), 8, xyz.size(), f) != xyz.size() || fread(ess.data(), 1, ess.size(), f) != ess.size()) return 1;
  fclose(f);
  const int ne = A.nrows, dim = 3;
  double t0 = now();
  const HostCsr Gb = palace::amg::DropRows(G, ess);
  std::vector<HostCsr> Pic(dim);
  for (int c = 0; c < dim; c++) Pic[c].nrows = ne, Pic[c].ncols = (int)nv, Pic[c].rowptr.assign((size_t)ne + 1, 0);
  for (int e = 0; e < ne; e++) {
    if (!ess[e])
      for (int c = 0; c < dim; c++) {
        double tc = 0.0;
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) tc += G.val[a] * xyz[(size_t)G.col[a] * dim + c];
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) Pic[c].col.push_back(G.col[a]), Pic[c].val.push_back(0.5 * std::abs(G.val[a]) * tc);
      }
    for (int c = 0; c < dim; c++) Pic[c].rowptr[e + 1] = (int)Pic[c].col.size();
  }
  double t1 = now();
  const HostCsr AG = galerkin(A, Gb);
  double t2 = now();
  HostCsr B;
  B.nrows = B.ncols = dim * (int)nv;
  B.rowptr.assign(1, 0);
  for (int c = 0; c < dim; c++) {
    const HostCsr M = galerkin(A, Pic[c]);
    for (int r = 0; r < nv; r++) {
      for (int a = M.rowptr[r]; a < M.rowptr[r + 1]; a++) B.col.push_back(c * (int)nv + M.col[a]), B.val.push_back(M.val[a]);
      B.rowptr.push_back((int)B.col.size());
    }
  }
  double t3 = now();
  const palace::amg::Hierarchy hg = palace::amg::Setup(AG);
  double t4 = now();
  const palace::amg::Hierarchy hp = palace::amg::Setup(B);
  double t5 = now();
  std::printf("transfers %.3f s, G^T A G %.3f s, 3 x Pi_c^T A Pi_c %.3f s, hierarchy of G^T A G (%zu levels) %.3f s, of the Pi block (%zu levels) %.3f s: total %.3f s\n",
              t1 - t0, t2 - t1, t3 - t2, hg.A.size(), t4 - t3, hp.A.size(), t5 - t4, t5 - t0);
  double cs = checksum(AG) + checksum(B);
  for (const auto &h : {&hg, &hp}) {
    for (const HostCsr &M : h->A) cs += checksum(M);
    for (const HostCsr &M : h->P) cs += checksum(M);
  }
  std::printf("levels:");
  for (const HostCsr &M : hg.A) std::printf(" %d", M.nrows);
  std::printf(" |");
  for (const HostCsr &M : hp.A) std::printf(" %d", M.nrows);
  std::printf("\nchecksum %.17g\n", cs);
  return 0;
}
