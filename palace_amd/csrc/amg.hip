// Smoothed-aggregation set-up on the host (see amg.hpp).  No device code in this file.
#include "amg.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <numeric>
#include <thread>

#include "pa_internal.hpp"

namespace palace::amg {

namespace {
// The set-up is row-parallel almost everywhere (sparse products, the power iteration's matrix-vector products).  Rows are dealt to
// the threads in fixed blocks and every row is computed exactly as the serial code does, so the hierarchy does not depend on the
// number of threads (PALACE_AMD_SETUP_THREADS; default: the hardware's, at most 16 -- each keeps a dense accumulator row).
constexpr int kRowBlock = 2048;
int setup_threads() {
  static const int n = [] {
    const char *e = std::getenv("PALACE_AMD_SETUP_THREADS");
    int v = e ? std::atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(v, 16));
  }();
  return n;
}
// fn(thread, block, first row, last row + 1) for every block of kRowBlock rows, blocks handed out in order
template <class F>
void for_row_blocks(int nrows, F &&fn) {
  const int nblocks = (nrows + kRowBlock - 1) / kRowBlock;
  const int nt = std::min(setup_threads(), std::max(1, nblocks));
  if (nt == 1) {
    for (int b = 0; b < nblocks; b++) fn(0, b, b * kRowBlock, std::min(nrows, (b + 1) * kRowBlock));
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  std::exception_ptr err;
  std::atomic<bool> failed{false};
  for (int t = 0; t < nt; t++)
    pool.emplace_back([&, t] {
      try {
        for (int b = next++; b < nblocks; b = next++) fn(t, b, b * kRowBlock, std::min(nrows, (b + 1) * kRowBlock));
      } catch (...) {
        if (!failed.exchange(true)) err = std::current_exception();
      }
    });
  for (std::thread &th : pool) th.join();
  if (failed) std::rethrow_exception(err);
}
}  // namespace

HostCsr Transpose(const HostCsr &A) {
  HostCsr T;
  T.nrows = A.ncols, T.ncols = A.nrows;
  T.rowptr.assign((size_t)T.nrows + 1, 0);
  for (int c : A.col) T.rowptr[(size_t)c + 1]++;
  for (int r = 0; r < T.nrows; r++) T.rowptr[r + 1] += T.rowptr[r];
  T.col.resize(A.col.size()), T.val.resize(A.val.size());
  std::vector<int> fill(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int r = 0; r < A.nrows; r++)  // rows visited in order: the columns of T come out sorted
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++) {
      const int k = fill[A.col[a]]++;
      T.col[k] = r, T.val[k] = A.val[a];
    }
  return T;
}

HostCsr Multiply(const HostCsr &A, const HostCsr &B) {
  PA_REQUIRE(A.ncols == B.nrows, "dimension mismatch in the sparse product");
  HostCsr C;
  C.nrows = A.nrows, C.ncols = B.ncols;
  C.rowptr.assign((size_t)C.nrows + 1, 0);
  // row by row with a dense accumulator per thread (Gustavson); the rows of a block go to the block's own arrays, which are
  // strung together afterwards
  const int nblocks = (A.nrows + kRowBlock - 1) / kRowBlock;
  struct Scratch {
    std::vector<double> acc, val;  // dense accumulator row; the entries of this thread's blocks, one after the other
    std::vector<int> mark, cols, col;
  };
  struct Piece {
    int thread = 0;
    size_t first = 0, count = 0;
  };
  std::vector<Scratch> scratch((size_t)setup_threads());
  std::vector<Piece> piece((size_t)nblocks);
  for_row_blocks(A.nrows, [&](int t, int blk, int r0, int r1) {
    Scratch &w = scratch[(size_t)t];
    if (w.mark.empty()) w.acc.assign((size_t)B.ncols, 0.0), w.mark.assign((size_t)B.ncols, -1);
    const size_t start = w.col.size();
    for (int r = r0; r < r1; r++) {
      w.cols.clear();
      for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++) {
        const int k = A.col[a];
        const double v = A.val[a];
        for (int b = B.rowptr[k]; b < B.rowptr[k + 1]; b++) {
          const int c = B.col[b];
          if (w.mark[c] != r) w.mark[c] = r, w.acc[c] = 0.0, w.cols.push_back(c);
          w.acc[c] += v * B.val[b];
        }
      }
      std::sort(w.cols.begin(), w.cols.end());
      for (int c : w.cols)
        if (w.acc[c] != 0.0) w.col.push_back(c), w.val.push_back(w.acc[c]);
      C.rowptr[r + 1] = (int)(w.col.size() - start);  // (within the block; made global below)
    }
    piece[(size_t)blk] = Piece{t, start, w.col.size() - start};
  });
  std::vector<long long> first((size_t)nblocks + 1, 0);
  for (int b = 0; b < nblocks; b++) first[b + 1] = first[b] + (long long)piece[(size_t)b].count;
  PA_REQUIRE(first[nblocks] < (1ll << 31), "sparse product too large (int32 CSR)");
  C.col.resize((size_t)first[nblocks]), C.val.resize((size_t)first[nblocks]);
  for_row_blocks(A.nrows, [&](int, int blk, int r0, int r1) {
    const Piece &p = piece[(size_t)blk];
    const Scratch &w = scratch[(size_t)p.thread];
    std::copy(w.col.begin() + (long)p.first, w.col.begin() + (long)(p.first + p.count), C.col.begin() + first[blk]);
    std::copy(w.val.begin() + (long)p.first, w.val.begin() + (long)(p.first + p.count), C.val.begin() + first[blk]);
    for (int r = r0; r < r1; r++) C.rowptr[r + 1] += (int)first[blk];
  });
  return C;
}

void Mult(const HostCsr &A, const std::vector<double> &x, std::vector<double> &y) {
  y.assign((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++) {
    double s = 0.0;
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++) s += A.val[a] * x[A.col[a]];
    y[r] = s;
  }
}

namespace {
std::vector<double> diagonal(const HostCsr &A) {
  std::vector<double> d((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++)
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++)
      if (A.col[a] == r) d[r] = A.val[a];
  return d;
}
inline bool strong(double aij, double dii, double djj, double theta) {
  return aij * aij >= theta * theta * std::abs(dii * djj);
}
// spectral radius of D_F^-1 A_F (filtered matrix of SmoothProlongator), a few power iterations from a fixed vector
double filtered_radius(const HostCsr &A, double theta) {
  const std::vector<double> d = diagonal(A);
  const int n = A.nrows;
  std::vector<double> dF((size_t)n), u((size_t)n), v((size_t)n);
  for (int i = 0; i < n; i++) {
    double dii = d[i];
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++)
      if (A.col[a] != i && !strong(A.val[a], d[i], d[A.col[a]], theta)) dii += A.val[a];
    if (!(std::abs(dii) > 0.1 * std::abs(d[i]))) dii = d[i];
    dF[i] = dii != 0.0 ? dii : 1.0;
    u[i] = 1.0 + 0.5 * std::sin(1.7 * i + 0.3);  // deterministic, not orthogonal to the top of the spectrum in practice
  }
  double rho = 0.0;
  for (int it = 0; it < 20; it++) {
    double nrm = 0.0;
    for (int i = 0; i < n; i++) nrm += u[i] * u[i];
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) break;
    for (int i = 0; i < n; i++) u[i] /= nrm;
    for_row_blocks(n, [&](int, int, int i0, int i1) {
      for (int i = i0; i < i1; i++) {
        double s = 0.0;
        for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) {
          const int j = A.col[a];
          if (j == i)
            s += dF[i] * u[i];
          else if (strong(A.val[a], d[i], d[j], theta))
            s += A.val[a] * u[j];
        }
        v[i] = s / dF[i];
      }
    });
    double r = 0.0;
    for (int i = 0; i < n; i++) r += v[i] * v[i];
    rho = std::sqrt(r);
    u.swap(v);
  }
  return rho;
}
}  // namespace

namespace {
std::vector<int> aggregate_impl(const HostCsr &A, double theta, int &num_aggregates, const int *blk);
}
std::vector<int> Aggregate(const HostCsr &A, double theta, int &num_aggregates) { return aggregate_impl(A, theta, num_aggregates, nullptr); }

std::vector<int> AggregateBlocks(const HostCsr &A, double theta, const std::vector<int> &block_off, int &num_aggregates,
                                 std::vector<int> &agg_off) {
  const int nb = (int)block_off.size() - 1;
  PA_REQUIRE(nb >= 1 && block_off.front() == 0 && block_off.back() == A.nrows, "row blocks do not cover the matrix");
  std::vector<int> blk((size_t)A.nrows);
  for (int b = 0; b < nb; b++)
    for (int i = block_off[(size_t)b]; i < block_off[(size_t)b + 1]; i++) blk[(size_t)i] = b;
  int na = 0;
  std::vector<int> agg = aggregate_impl(A, theta, na, blk.data());
  // aggregates block by block, in the order of their first row inside a block
  std::vector<int> ablk((size_t)na, -1), renum((size_t)na, -1);
  for (int i = 0; i < A.nrows; i++)
    if (agg[(size_t)i] >= 0 && ablk[(size_t)agg[(size_t)i]] < 0) ablk[(size_t)agg[(size_t)i]] = blk[(size_t)i];
  agg_off.assign((size_t)nb + 1, 0);
  for (int a = 0; a < na; a++) agg_off[(size_t)ablk[(size_t)a] + 1]++;
  for (int b = 0; b < nb; b++) agg_off[(size_t)b + 1] += agg_off[(size_t)b];
  std::vector<int> next(agg_off.begin(), agg_off.end() - 1);
  for (int i = 0; i < A.nrows; i++) {
    const int a = agg[(size_t)i];
    if (a >= 0 && renum[(size_t)a] < 0) renum[(size_t)a] = next[(size_t)ablk[(size_t)a]]++;
  }
  for (int i = 0; i < A.nrows; i++)
    if (agg[(size_t)i] >= 0) agg[(size_t)i] = renum[(size_t)agg[(size_t)i]];
  num_aggregates = na;
  return agg;
}

namespace {
std::vector<int> aggregate_impl(const HostCsr &A, double theta, int &num_aggregates, const int *blk) {
  PA_REQUIRE(A.nrows == A.ncols, "aggregation needs a square matrix");
  const int n = A.nrows;
  const std::vector<double> d = diagonal(A);
  // (blk: ties between rows of different blocks do not count)
  auto strong = [&](double aij, double dii, double djj, double th, int i, int j) {
    return (!blk || blk[i] == blk[j]) && palace::amg::strong(aij, dii, djj, th);
  };
  std::vector<int> agg((size_t)n, -1);
  // rows without a strong off-diagonal entry (eliminated essential dofs: a lone diagonal) stay out of the coarse problem:
  // the smoother solves them, and carried along they would form one aggregate each and stop the coarsening
  // (decided with the UNMASKED strength test: a row whose strong neighbours all belong to other blocks -- an interface row of a
  // thin partition -- is not isolated; it founds or joins an aggregate of its own block in passes 1-3, a singleton at worst, so the
  // coarse space does not lose rows because of where the partition cuts.  Round-5 advisor finding.)
  std::vector<char> isolated((size_t)n, 1);
  for (int i = 0; i < n; i++)
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1] && isolated[i]; a++)
      if (A.col[a] != i && palace::amg::strong(A.val[a], d[i], d[A.col[a]], theta)) isolated[i] = 0;
  int na = 0;
  // pass 1: a node whose strong neighbours are all free founds an aggregate with them
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0 || isolated[i]) continue;
    bool free_nbrs = true, any = false;
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1] && free_nbrs; a++) {
      const int j = A.col[a];
      if (j == i || !strong(A.val[a], d[i], d[j], theta, i, j)) continue;
      any = true;
      if (agg[j] >= 0) free_nbrs = false;
    }
    if (!free_nbrs || !any) continue;
    agg[i] = na;
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) {
      const int j = A.col[a];
      if (j != i && strong(A.val[a], d[i], d[j], theta, i, j)) agg[j] = na;
    }
    na++;
  }
  // pass 2: the rest joins the aggregate (as formed in pass 1) it is most strongly tied to
  const std::vector<int> pass1(agg);
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0 || isolated[i]) continue;
    double best = 0.0;
    int to = -1;
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) {
      const int j = A.col[a];
      if (j == i || pass1[j] < 0 || !strong(A.val[a], d[i], d[j], theta, i, j)) continue;
      if (std::abs(A.val[a]) > best) best = std::abs(A.val[a]), to = pass1[j];
    }
    if (to >= 0) agg[i] = to;
  }
  // pass 3: what is still free (no strong tie to any aggregate) forms aggregates with its free strong neighbours
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0 || isolated[i]) continue;
    agg[i] = na;
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) {
      const int j = A.col[a];
      if (j != i && agg[j] < 0 && !isolated[j] && strong(A.val[a], d[i], d[j], theta, i, j)) agg[j] = na;
    }
    na++;
  }
  num_aggregates = na;
  return agg;
}
}  // namespace

HostCsr TentativeProlongator(const std::vector<int> &aggregate, int num_aggregates) {
  HostCsr T;
  T.nrows = (int)aggregate.size(), T.ncols = num_aggregates;
  std::vector<int> size((size_t)num_aggregates, 0);
  for (int a : aggregate)
    if (a >= 0) size[a]++;
  T.rowptr.assign((size_t)T.nrows + 1, 0);
  for (size_t i = 0; i < aggregate.size(); i++) {  // (rows outside every aggregate stay empty)
    if (aggregate[i] >= 0) T.col.push_back(aggregate[i]), T.val.push_back(1.0 / std::sqrt((double)size[aggregate[i]]));
    T.rowptr[i + 1] = (int)T.col.size();
  }
  return T;
}

HostCsr SmoothProlongator(const HostCsr &A, const HostCsr &T, double theta, double omega) {
  // omega = 4 / (3 rho(D_F^-1 A_F)) unless given (rho padded by 10 %: the power iteration approaches it from below)
  if (omega <= 0.0) omega = 4.0 / (3.0 * std::max(1.0, 1.1 * filtered_radius(A, theta)));
  // filtered matrix: weak off-diagonal entries are dropped and added to the diagonal (row sums kept)
  const std::vector<double> d = diagonal(A);
  HostCsr S;  // S = I - omega D_F^-1 A_F
  S.nrows = S.ncols = A.nrows;
  S.rowptr.assign((size_t)A.nrows + 1, 0);
  for (int i = 0; i < A.nrows; i++) {
    double dii = d[i];
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++)
      if (A.col[a] != i && !strong(A.val[a], d[i], d[A.col[a]], theta)) dii += A.val[a];
    if (!(std::abs(dii) > 0.1 * std::abs(d[i]))) dii = d[i];  // lumping must not cancel the diagonal
    if (dii == 0.0) {  // an empty row: nothing to smooth, the row of P stays what T has (nothing, for an isolated dof)
      S.col.push_back(i), S.val.push_back(1.0);
      S.rowptr[i + 1] = (int)S.col.size();
      continue;
    }
    for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) {
      const int j = A.col[a];
      if (j == i)
        S.col.push_back(j), S.val.push_back(1.0 - omega);
      else if (strong(A.val[a], d[i], d[j], theta))
        S.col.push_back(j), S.val.push_back(-omega * A.val[a] / dii);
    }
    S.rowptr[i + 1] = (int)S.col.size();
  }
  return Multiply(S, T);
}

HostCsr DropRows(const HostCsr &A, const std::vector<char> &flag) {
  PA_REQUIRE((int)flag.size() == A.nrows, "flag size mismatch");
  HostCsr B;
  B.nrows = A.nrows, B.ncols = A.ncols;
  B.rowptr.assign((size_t)A.nrows + 1, 0);
  for (int i = 0; i < A.nrows; i++) {
    if (!flag[i])
      for (int a = A.rowptr[i]; a < A.rowptr[i + 1]; a++) B.col.push_back(A.col[a]), B.val.push_back(A.val[a]);
    B.rowptr[i + 1] = (int)B.col.size();
  }
  return B;
}


Hierarchy SetupBlocks(const HostCsr &A0, const std::vector<int> &block_off, std::vector<std::vector<int>> &level_off, int max_levels,
                      int coarse_size, double theta, double omega) {
  Hierarchy h;
  h.A.push_back(A0);
  level_off.assign(1, block_off);
  while ((int)h.A.size() < max_levels && h.A.back().nrows > coarse_size) {
    const HostCsr &A = h.A.back();
    int na = 0;
    std::vector<int> next_off;
    const std::vector<int> agg = AggregateBlocks(A, theta, level_off.back(), na, next_off);
    if (na == 0 || na >= A.nrows) break;  // no coarsening left
    HostCsr P = SmoothProlongator(A, TentativeProlongator(agg, na), theta, omega);
    HostCsr Ac = Multiply(Transpose(P), Multiply(A, P));
    h.P.push_back(std::move(P));
    h.A.push_back(std::move(Ac));
    level_off.push_back(std::move(next_off));
  }
  return h;
}

Hierarchy Setup(const HostCsr &A0, int max_levels, int coarse_size, double theta, double omega) {
  Hierarchy h;
  h.A.push_back(A0);
  while ((int)h.A.size() < max_levels && h.A.back().nrows > coarse_size) {
    const HostCsr &A = h.A.back();
    int na = 0;
    const std::vector<int> agg = Aggregate(A, theta, na);
    if (na == 0 || na >= A.nrows) break;  // no coarsening left
    HostCsr P = SmoothProlongator(A, TentativeProlongator(agg, na), theta, omega);
    HostCsr Ac = Multiply(Transpose(P), Multiply(A, P));
    h.P.push_back(std::move(P));
    h.A.push_back(std::move(Ac));
  }
  return h;
}

}  // namespace palace::amg
