// Device side of the native coarse-level solvers (see amg_solver.hpp): V-cycles of sparse matrix-vector products and fused
// vector kernels on the context's stream.
#include "amg_solver.hpp"

#include "amg_dist.hpp"

#include <algorithm>
#include <cmath>

namespace palace {

// ---- matrices ---------------------------------------------------------------------------------------------------------------
DeviceCsr::DeviceCsr(const Context &ctx, const amg::HostCsr &h, bool symmetric) {
  PA_REQUIRE((int)h.rowptr.size() == h.nrows + 1, "malformed CSR matrix");
  m_.symmetric = symmetric;
  m_.nrows = h.nrows, m_.ncols = (h.ncols == h.nrows) ? 0 : h.ncols, m_.nnz = h.nnz();
  std::vector<int32_t> rp(h.rowptr.begin(), h.rowptr.end()), cl(h.col.begin(), h.col.end());
  m_.d_rowptr = pa::dev_upload(rp.data(), rp.size(), ctx.stream);
  m_.d_col = pa::dev_upload(cl.data(), cl.size(), ctx.stream);
  m_.d_val = pa::dev_upload(h.val.data(), h.val.size(), ctx.stream);
  op_ = std::make_unique<CsrOperator>(ctx, &m_);
}
DeviceCsr::~DeviceCsr() {
  op_.reset();
  (void)hipFree(m_.d_rowptr), (void)hipFree(m_.d_col), (void)hipFree(m_.d_val);
}

amg::HostCsr DownloadCsr(const pa_csr &m, const int32_t *ess, int n_ess) {
  amg::HostCsr h;
  h.nrows = m.nrows, h.ncols = m.ncols ? m.ncols : m.nrows;
  std::vector<int32_t> rp((size_t)m.nrows + 1), cl((size_t)m.nnz);
  h.val.resize((size_t)m.nnz);
  PA_HIP(hipMemcpy(rp.data(), m.d_rowptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  PA_HIP(hipMemcpy(cl.data(), m.d_col, cl.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  PA_HIP(hipMemcpy(h.val.data(), m.d_val, h.val.size() * sizeof(double), hipMemcpyDeviceToHost));
  h.rowptr.assign(rp.begin(), rp.end());
  h.col.assign(cl.begin(), cl.end());
  if (n_ess > 0) {  // rows and columns of the essential dofs zeroed, diagonal 1 (rap.cpp:131-149)
    PA_REQUIRE(h.nrows == h.ncols, "essential dofs on a rectangular matrix");
    std::vector<char> flag((size_t)h.nrows, 0);
    for (int i = 0; i < n_ess; i++) {
      PA_REQUIRE(ess[i] >= 0 && ess[i] < h.nrows, "essential dof out of range");
      flag[ess[i]] = 1;
    }
    amg::HostCsr e;
    e.nrows = h.nrows, e.ncols = h.ncols;
    e.rowptr.assign((size_t)h.nrows + 1, 0);
    for (int r = 0; r < h.nrows; r++) {
      if (flag[r]) {
        e.col.push_back(r), e.val.push_back(1.0);
      } else {
        for (int a = h.rowptr[r]; a < h.rowptr[r + 1]; a++)
          if (!flag[h.col[a]]) e.col.push_back(h.col[a]), e.val.push_back(h.val[a]);
      }
      e.rowptr[r + 1] = (int)e.col.size();
    }
    return e;
  }
  return h;
}

namespace {

// 1 / sum_j |a_ij| (empty rows: 0, such dofs are never corrected)
std::vector<double> l1_inverse(const amg::HostCsr &A) {
  std::vector<double> d((size_t)A.nrows, 0.0);
  for (int r = 0; r < A.nrows; r++) {
    double s = 0.0;
    for (int a = A.rowptr[r]; a < A.rowptr[r + 1]; a++) s += std::abs(A.val[a]);
    d[r] = s > 0.0 ? 1.0 / s : 0.0;
  }
  return d;
}

Vector upload(const Context &ctx, const std::vector<double> &v) {
  Vector out((int)v.size());
  if (!v.empty()) PA_HIP(hipMemcpyAsync(out.Data(), v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, ctx.stream));
  PA_HIP(hipStreamSynchronize(ctx.stream));
  return out;
}

// Pseudo-inverse of a small symmetric positive semi-definite matrix (row-major n x n): Cholesky when it goes through,
// else the eigen-decomposition by cyclic Jacobi rotations with the eigenvalues below 1e-12 of the largest one dropped (a
// singular last level: the gradient-space matrix of a problem without a mass term, a pure Neumann block).
std::vector<double> pseudo_inverse(std::vector<double> M, int n) {
  std::vector<double> inv((size_t)n * n, 0.0);
  if (n == 0) return inv;
  double dmax = 0.0;
  for (int i = 0; i < n; i++) dmax = std::max(dmax, std::abs(M[(size_t)i * n + i]));
  {  // Cholesky M = L L^T
    std::vector<double> L(M);
    bool ok = dmax > 0.0;
    for (int j = 0; j < n && ok; j++) {
      double s = L[(size_t)j * n + j];
      for (int k = 0; k < j; k++) s -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
      if (!(s > 1e-12 * dmax)) {
        ok = false;
        break;
      }
      const double ljj = std::sqrt(s);
      L[(size_t)j * n + j] = ljj;
      for (int i = j + 1; i < n; i++) {
        double t = L[(size_t)i * n + j];
        for (int k = 0; k < j; k++) t -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
        L[(size_t)i * n + j] = t / ljj;
      }
    }
    if (ok) {
      // columns of the inverse: L L^T x = e_c
      std::vector<double> y((size_t)n);
      for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) {
          double t = (i == c) ? 1.0 : 0.0;
          for (int k = 0; k < i; k++) t -= L[(size_t)i * n + k] * y[k];
          y[i] = t / L[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
          double t = y[i];
          for (int k = i + 1; k < n; k++) t -= L[(size_t)k * n + i] * inv[(size_t)k * n + c];
          inv[(size_t)i * n + c] = t / L[(size_t)i * n + i];
        }
      }
      return inv;
    }
  }
  // cyclic Jacobi: M -> diagonal, V accumulates the rotations
  std::vector<double> V((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) off += M[(size_t)p * n + q] * M[(size_t)p * n + q];
    if (off <= 1e-30 * dmax * dmax * n * n) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = M[(size_t)p * n + q];
        if (std::abs(apq) <= 1e-300) continue;
        const double tau = (M[(size_t)q * n + q] - M[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::abs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; k++) {  // columns p, q
          const double mkp = M[(size_t)k * n + p], mkq = M[(size_t)k * n + q];
          M[(size_t)k * n + p] = c * mkp - s * mkq, M[(size_t)k * n + q] = s * mkp + c * mkq;
        }
        for (int k = 0; k < n; k++) {  // rows p, q
          const double mpk = M[(size_t)p * n + k], mqk = M[(size_t)q * n + k];
          M[(size_t)p * n + k] = c * mpk - s * mqk, M[(size_t)q * n + k] = s * mpk + c * mqk;
        }
        for (int k = 0; k < n; k++) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq, V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  double lmax = 0.0;
  for (int i = 0; i < n; i++) lmax = std::max(lmax, M[(size_t)i * n + i]);
  for (int e = 0; e < n; e++) {
    const double lam = M[(size_t)e * n + e];
    if (!(lam > 1e-12 * lmax)) continue;
    for (int i = 0; i < n; i++) {
      const double f = V[(size_t)i * n + e] / lam;
      for (int j = 0; j < n; j++) inv[(size_t)i * n + j] += f * V[(size_t)j * n + e];
    }
  }
  return inv;
}

amg::HostCsr dense_to_csr(const std::vector<double> &M, int n) {
  amg::HostCsr h;
  h.nrows = h.ncols = n;
  h.rowptr.resize((size_t)n + 1);
  h.col.resize((size_t)n * n), h.val = M;
  for (int r = 0; r <= n; r++) h.rowptr[r] = r * n;
  for (int r = 0; r < n; r++)
    for (int c = 0; c < n; c++) h.col[(size_t)r * n + c] = c;
  return h;
}

// 4th-kind Chebyshev smoothing of order k on D^-1 A with lambda_max = 1 (chebyshev.cpp:190-220 with the eigenvalue bound of
// the l1 scaling): x <- x + p(D^-1 A) D^-1 (b - A x).  r, d, t: work vectors of the size of x.
void cheb4(const Context &c, const Operator &A, const Vector &dinv, int order, const Vector &b, Vector &x, bool zero_guess,
           Vector &r, Vector &d, Vector &t, bool fused = false) {
  if (fused) {
    // Round 6: the same polynomial in its accumulated form (linalg.hip: ChebyshevSmoother, e_k = d_0 + ... + d_{k-1}) with every
    // product consumed in the sparse product's epilogue (CsrOperator::MultResidual / MultChebyStep): order 2 is TWO launches
    // instead of six.  e_{k+1} = e_k + sd (e_k - e_{k-1}) + sr D^-1 (r_0 - A e_k); r_0 is the right-hand side itself with a zero guess.
    const Vector *r0 = &b;
    if (zero_guess) {
      linalg::ChebyOrder0(c, 4.0 / 3.0, dinv, b, order > 1 ? d : x);  // e_1 (order 1: the result)
    } else {
      A.MultResidual(x, b, &r, &dinv, 4.0 / 3.0, &d);  // r_0 = b - A x, e_1 = c_0 D^-1 r_0
      r0 = &r;
      if (order <= 1) linalg::AXPY(c, 1.0, d, x);
    }
    Vector *ek = &d, *ep = &t;
    for (int k = 1; k < order; k++) {
      const double sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / (2.0 * k + 3.0);
      const bool last = k == order - 1;
      A.MultChebyStep(*ek, Operator::ChebyStepArgs{sd, sr, &dinv, r0, k == 1 ? nullptr : ep, last ? &x : ep, last && !zero_guess});
      if (!last) std::swap(ek, ep);
    }
    return;
  }
  if (zero_guess) {
    linalg::Copy(c, b, r);
    linalg::Fill(c, x, 0.0);
  } else {
    A.Mult(x, r);
    linalg::AXPBY(c, 1.0, b, -1.0, r);
  }
  linalg::ChebyOrder0(c, 4.0 / 3.0, dinv, r, d);
  for (int k = 1; k < order; k++) {
    const double sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / (2.0 * k + 3.0);
    A.Mult(d, t);
    linalg::ChebyStep(c, sd, sr, dinv, t, r, d, x);  // x += d; r -= A d; d = sd d + sr D^-1 r
  }
  linalg::AXPY(c, 1.0, d, x);
}

}  // namespace

// ---- AMG --------------------------------------------------------------------------------------------------------------------
AmgSolver::AmgSolver(const Context &ctx, const amg::HostCsr &A, const AmgOptions &opt) : ctx_(&ctx), opt_(opt) {
  PA_REQUIRE(A.nrows == A.ncols, "AmgSolver needs a square matrix");
  height = width = A.nrows;
  host_ = amg::Setup(A, opt.max_levels, opt.coarse_size, opt.theta);
  const size_t nl = host_.A.size();
  lv_.resize(nl);
  for (size_t l = 0; l < nl; l++) {
    Level &L = lv_[l];
    const amg::HostCsr &Al = host_.A[l];
    L.A = std::make_unique<DeviceCsr>(ctx, Al, true);
    const int n = Al.nrows;
    L.r.SetSize(n), L.d.SetSize(n), L.t.SetSize(n);
    if (l > 0) L.x.SetSize(n), L.b.SetSize(n);
    if (l + 1 < nl) {
      L.dinv = upload(ctx, l1_inverse(Al));
      L.P = std::make_unique<DeviceCsr>(ctx, host_.P[l], false);
      L.R = std::make_unique<DeviceCsr>(ctx, amg::Transpose(host_.P[l]), false);
    }
  }
  // last level: dense pseudo-inverse (a level that stopped coarsening above the direct-solve size keeps a smoother instead)
  const amg::HostCsr &Ac = host_.A.back();
  if (Ac.nrows <= std::max(opt.coarse_size, 1) * 4 && Ac.nrows <= 2000) {
    std::vector<double> M((size_t)Ac.nrows * Ac.nrows, 0.0);
    for (int r = 0; r < Ac.nrows; r++)
      for (int a = Ac.rowptr[r]; a < Ac.rowptr[r + 1]; a++) M[(size_t)r * Ac.nrows + Ac.col[a]] = Ac.val[a];
    host_cinv_ = pseudo_inverse(std::move(M), Ac.nrows);
    Cinv_ = std::make_unique<DeviceCsr>(ctx, dense_to_csr(host_cinv_, Ac.nrows), true);
  } else {
    lv_.back().dinv = upload(ctx, l1_inverse(Ac));
  }
  fused_ = !lv_.empty() && lv_[0].A->Op().PrepareChebyStep();
}

void AmgSolver::Smooth(const Level &L, const Vector &b, Vector &x, bool zero_guess) const {
  cheb4(*ctx_, L.A->Op(), L.dinv, opt_.smooth_order, b, x, zero_guess, L.r, L.d, L.t, fused_);
}

void AmgSolver::Cycle(size_t l, const Vector &b, Vector &x) const {
  const Context &c = *ctx_;
  const Level &L = lv_[l];
  if (l + 1 == lv_.size()) {
    if (Cinv_) {
      Cinv_->Op().Mult(b, x);
    } else {  // no direct solve available: a few smoothing steps
      Smooth(L, b, x, true);
      for (int it = 0; it < 3; it++) Smooth(L, b, x, false);
    }
    return;
  }
  Smooth(L, b, x, true);
  if (fused_) {
    L.A->Op().MultResidual(x, b, &L.r);  // r = b - A x in the product's epilogue
  } else {
    L.A->Op().Mult(x, L.r);
    linalg::AXPBY(c, 1.0, b, -1.0, L.r);
  }
  const Level &N = lv_[l + 1];
  L.R->Op().Mult(L.r, N.b);
  Cycle(l + 1, N.b, N.x);
  L.P->Op().AddMult(N.x, x, 1.0);
  Smooth(L, b, x, false);
}

void AmgSolver::Mult(const Vector &b, Vector &x) const {
  PA_REQUIRE(b.Size() == height && x.Size() == height, "size mismatch in AmgSolver");
  if (!height) return;
  Cycle(0, b, x);
}

// ---- AMS --------------------------------------------------------------------------------------------------------------------
namespace {
// T^T A T with empty rows / columns given a unit diagonal (vertices whose edges are all essential)
amg::HostCsr galerkin(const amg::HostCsr &A, const amg::HostCsr &T) {
  amg::HostCsr M = amg::Multiply(amg::Transpose(T), amg::Multiply(A, T));
  amg::HostCsr out;
  out.nrows = out.ncols = M.nrows;
  out.rowptr.assign((size_t)M.nrows + 1, 0);
  for (int r = 0; r < M.nrows; r++) {
    bool has_diag = false;
    for (int a = M.rowptr[r]; a < M.rowptr[r + 1]; a++) has_diag = has_diag || (M.col[a] == r && M.val[a] != 0.0);
    if (!has_diag) {  // (a zero diagonal of a positive semi-definite matrix means a zero row)
      out.col.push_back(r), out.val.push_back(1.0);
    } else {
      for (int a = M.rowptr[r]; a < M.rowptr[r + 1]; a++) out.col.push_back(M.col[a]), out.val.push_back(M.val[a]);
    }
    out.rowptr[r + 1] = (int)out.col.size();
  }
  return out;
}
}  // namespace

namespace {
// Host side of the AMS set-up, shared by the one-rank / replicated solver and the distributed one: the transfers without the
// essential edges and the auxiliary matrices.  voff [ranks + 1]: ownership ranges of the vertices; column (c, v) of Pi is numbered
// dim voff[r] + c nv_r + (v - voff[r]) for a vertex v of rank r -- rank by rank, and component by component inside a rank (one rank:
// c nv + v, HYPRE's layout of the three coordinate blocks) -- so that the rows of the nodal auxiliary problem a rank owns are
// consecutive too.
struct AmsHost {
  amg::HostCsr Gb, Pi, BG, BPi;
  std::vector<int> woff;  // ownership ranges of the columns of Pi (dim voff)
};
AmsHost ams_host(const amg::HostCsr &A, const amg::HostCsr &G, const double *coords, int dim, const std::vector<char> &ess_flag,
                 bool singular, const std::vector<int> &voff) {
  PA_REQUIRE(A.nrows == A.ncols && G.nrows == A.nrows, "AmsSolver: A [edges x edges] and G [edges x vertices] expected");
  PA_REQUIRE(dim >= 2 && dim <= 3 && coords, "AmsSolver needs the vertex coordinates");
  PA_REQUIRE((int)ess_flag.size() == A.nrows, "essential flag size mismatch");
  const int ne = A.nrows, nv = G.ncols, nr = (int)voff.size() - 1;
  PA_REQUIRE(nr >= 1 && voff.front() == 0 && voff.back() == nv, "AmsSolver: vertex ranges");
  AmsHost h;
  std::vector<int> base((size_t)nv), stride((size_t)nv);  // column of (c, v) = base[v] + c stride[v]
  for (int r = 0; r < nr; r++)
    for (int v = voff[(size_t)r]; v < voff[(size_t)r + 1]; v++)
      base[(size_t)v] = dim * voff[(size_t)r] + (v - voff[(size_t)r]), stride[(size_t)v] = voff[(size_t)r + 1] - voff[(size_t)r];
  h.woff.resize(voff.size());
  for (size_t r = 0; r < voff.size(); r++) h.woff[r] = dim * voff[r];
  // transfers without the essential edges: no correction ever touches those
  h.Gb = amg::DropRows(G, ess_flag);
  // Pi_c = |G| diag(G x_c) / 2 (HYPRE_AMSSetCoordinateVectors; ams.cpp:64-100)
  amg::HostCsr &Pi = h.Pi;
  Pi.nrows = ne, Pi.ncols = dim * nv;
  Pi.rowptr.assign((size_t)ne + 1, 0);
  std::vector<amg::HostCsr> Pic((size_t)dim);
  for (int c = 0; c < dim; c++) Pic[c].nrows = ne, Pic[c].ncols = nv, Pic[c].rowptr.assign((size_t)ne + 1, 0);
  std::vector<std::pair<int, double>> row;
  for (int e = 0; e < ne; e++) {
    if (!ess_flag[e]) {
      row.clear();
      for (int c = 0; c < dim; c++) {
        double tc = 0.0;
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) tc += G.val[a] * coords[(size_t)G.col[a] * dim + c];
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) {
          const double v = 0.5 * std::abs(G.val[a]) * tc;
          row.emplace_back(base[(size_t)G.col[a]] + c * stride[(size_t)G.col[a]], v);
          Pic[c].col.push_back(G.col[a]), Pic[c].val.push_back(v);
        }
      }
      std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
      for (const auto &t : row) Pi.col.push_back(t.first), Pi.val.push_back(t.second);
    }
    Pi.rowptr[e + 1] = (int)Pi.col.size();
    for (int c = 0; c < dim; c++) Pic[c].rowptr[e + 1] = (int)Pic[c].col.size();
  }
  // auxiliary matrices: G^T A G and the block-diagonal matrix of the Pi_c^T A Pi_c (one hierarchy serves the three additive
  // scalar corrections: its aggregates never cross the blocks)
  if (!singular) h.BG = galerkin(A, h.Gb);
  {
    std::vector<amg::HostCsr> M((size_t)dim);
    for (int c = 0; c < dim; c++) M[c] = galerkin(A, Pic[c]);
    amg::HostCsr &B = h.BPi;
    B.nrows = B.ncols = dim * nv;
    B.rowptr.assign(1, 0);
    for (int r = 0; r < nr; r++)
      for (int c = 0; c < dim; c++)
        for (int v = voff[(size_t)r]; v < voff[(size_t)r + 1]; v++) {  // row base[v] + c stride[v]: increasing in this order
          for (int a = M[c].rowptr[v]; a < M[c].rowptr[v + 1]; a++) {
            const int w = M[c].col[a];
            B.col.push_back(base[(size_t)w] + c * stride[(size_t)w]), B.val.push_back(M[c].val[a]);
          }
          B.rowptr.push_back((int)B.col.size());
        }
  }
  return h;
}
}  // namespace

AmsSolver::AmsSolver(const Context &ctx, const amg::HostCsr &A, const amg::HostCsr &G, const double *coords, int dim,
                     const std::vector<char> &ess_flag, const AmsOptions &opt)
    : ctx_(&ctx), opt_(opt) {
  height = width = A.nrows;
  const int ne = A.nrows, nv = G.ncols;
  const AmsHost h = ams_host(A, G, coords, dim, ess_flag, opt.singular, std::vector<int>{0, nv});
  if (!opt.singular) BG_ = std::make_unique<AmgSolver>(ctx, h.BG, opt.amg);
  BPi_ = std::make_unique<AmgSolver>(ctx, h.BPi, opt.amg);
  A_ = std::make_unique<DeviceCsr>(ctx, A, true);
  if (!opt.singular) {
    G_ = std::make_unique<DeviceCsr>(ctx, h.Gb, false);
    Gt_ = std::make_unique<DeviceCsr>(ctx, amg::Transpose(h.Gb), false);
    bg_.SetSize(nv), xg_.SetSize(nv);
  }
  Pi_ = std::make_unique<DeviceCsr>(ctx, h.Pi, false);
  Pit_ = std::make_unique<DeviceCsr>(ctx, amg::Transpose(h.Pi), false);
  bp_.SetSize(dim * nv), xp_.SetSize(dim * nv);
  dinv_ = upload(ctx, l1_inverse(A));
  r_.SetSize(ne), d_.SetSize(ne), t_.SetSize(ne);
  fused_ = A_->Op().PrepareChebyStep();
}

void AmsSolver::Smooth(const Vector &b, Vector &x, bool zero_guess) const {
  cheb4(*ctx_, A_->Op(), dinv_, opt_.smooth_order, b, x, zero_guess, r_, d_, t_, fused_);
}

// x += T B T^T (b - A x)
void AmsSolver::Correct(const DeviceCsr &T, const DeviceCsr &Tt, const AmgSolver &B, const Vector &b, Vector &x, Vector &bc,
                        Vector &xc) const {
  if (fused_) {
    A_->Op().MultResidual(x, b, &r_);
  } else {
    A_->Op().Mult(x, r_);
    linalg::AXPBY(*ctx_, 1.0, b, -1.0, r_);
  }
  Tt.Op().Mult(r_, bc);
  B.Mult(bc, xc);
  T.Op().AddMult(xc, x, 1.0);
}

void AmsSolver::Mult(const Vector &b, Vector &x) const {
  PA_REQUIRE(b.Size() == height && x.Size() == height, "size mismatch in AmsSolver");
  if (!height) return;
  for (int it = 0; it < opt_.cycle_it; it++) {
    Smooth(b, x, it == 0 && !initial_guess);
    if (BG_) Correct(*G_, *Gt_, *BG_, b, x, bg_, xg_);
    Correct(*Pi_, *Pit_, *BPi_, b, x, bp_, xp_);
    if (BG_) Correct(*G_, *Gt_, *BG_, b, x, bg_, xg_);
    Smooth(b, x, false);
  }
}

// ---- the V-cycle distributed over the ranks (amg_dist.hpp) ---------------------------------------------------------------------

void DistSpace::Need(const amg::HostCsr &M, const std::vector<int> &row_off) {
  PA_REQUIRE((int)row_off.size() == size_ + 1 && row_off.back() == M.nrows && M.ncols == off_.back(), "distributed level: matrix / ranges");
  for (int s = 0; s < size_; s++) {
    const int lo = off_[(size_t)s], hi = off_[(size_t)s + 1];
    std::vector<int> &need = need_[(size_t)s];
    for (int a = M.rowptr[(size_t)row_off[(size_t)s]]; a < M.rowptr[(size_t)row_off[(size_t)s + 1]]; a++) {
      const int c = M.col[(size_t)a];
      if (c < lo || c >= hi) need.push_back(c);
    }
  }
}

void DistSpace::Finalize(Comm *comm) {
  long long total = 0;
  for (std::vector<int> &need : need_) {
    std::sort(need.begin(), need.end());
    need.erase(std::unique(need.begin(), need.end()), need.end());
    total += (long long)need.size();
  }
  ghosts_ = need_[(size_t)rank_];
  const int lo = off_[(size_t)rank_], n_own = NumOwned();
  local_of_.assign((size_t)off_.back(), -1);
  for (int i = 0; i < n_own; i++) local_of_[(size_t)(lo + i)] = i;
  for (size_t k = 0; k < ghosts_.size(); k++) local_of_[(size_t)ghosts_[k]] = n_own + (int)k;
  if (!comm || size_ == 1 || total == 0) return;  // (no coupling across ranks on this level anywhere: every rank decides the same)
  std::vector<int> nbr, soff, roff;
  std::vector<int32_t> sidx, ridx;
  Plan(nbr, soff, sidx, roff, ridx);
  halo_ = std::make_unique<Halo>(*comm, (int)nbr.size(), nbr.data(), soff.data(), sidx.data(), roff.data(), ridx.data());
  halo_->Validate(n_own, NumLocal());
}

void DistSpace::Plan(std::vector<int> &nbr, std::vector<int> &soff, std::vector<int32_t> &sidx, std::vector<int> &roff,
                     std::vector<int32_t> &ridx) const {
  PA_REQUIRE((int)need_.size() == size_, "distributed level: Plan after Release");
  // what I receive: my ghosts by owner (ascending global numbers = rank by rank = contiguous pieces of the ghost tail);
  // what I send: the entries of my range in the other ranks' lists
  const int lo = off_[(size_t)rank_], n_own = NumOwned();
  nbr.clear(), sidx.clear(), ridx.clear();
  soff.assign(1, 0), roff.assign(1, 0);
  for (int s = 0; s < size_; s++) {
    if (s == rank_) continue;
    const auto g0 = std::lower_bound(ghosts_.begin(), ghosts_.end(), off_[(size_t)s]);
    const auto g1 = std::lower_bound(ghosts_.begin(), ghosts_.end(), off_[(size_t)s + 1]);
    const std::vector<int> &theirs = need_[(size_t)s];
    const auto t0 = std::lower_bound(theirs.begin(), theirs.end(), lo), t1 = std::lower_bound(theirs.begin(), theirs.end(), lo + n_own);
    if (g0 == g1 && t0 == t1) continue;
    nbr.push_back(s);
    for (auto it = t0; it != t1; ++it) sidx.push_back((int32_t)(*it - lo));
    for (auto it = g0; it != g1; ++it) ridx.push_back((int32_t)(n_own + (int)(it - ghosts_.begin())));
    soff.push_back((int)sidx.size()), roff.push_back((int)ridx.size());
  }
}

void DistSpace::Release() {
  std::vector<std::vector<int>>().swap(need_);
  std::vector<int>().swap(local_of_);
}

amg::HostCsr DistSpace::Localize(const amg::HostCsr &M, const std::vector<int> &row_off) const {
  PA_REQUIRE(!local_of_.empty() || off_.back() == 0, "distributed level: Localize after Release");
  const int r0 = row_off[(size_t)rank_], r1 = row_off[(size_t)rank_ + 1];
  amg::HostCsr L;
  L.nrows = r1 - r0, L.ncols = NumLocal();
  L.rowptr.assign((size_t)L.nrows + 1, 0);
  const size_t nnz = (size_t)(M.rowptr[(size_t)r1] - M.rowptr[(size_t)r0]);
  L.col.reserve(nnz), L.val.reserve(nnz);
  std::vector<std::pair<int, double>> row;
  for (int r = r0; r < r1; r++) {
    row.clear();
    for (int a = M.rowptr[(size_t)r]; a < M.rowptr[(size_t)r + 1]; a++) {
      const int c = local_of_[(size_t)M.col[(size_t)a]];
      PA_REQUIRE(c >= 0, "distributed level: a column that was not announced (DistSpace::Need)");
      row.emplace_back(c, M.val[(size_t)a]);
    }
    std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    for (const auto &e : row) L.col.push_back(e.first), L.val.push_back(e.second);
    L.rowptr[(size_t)(r - r0) + 1] = (int)L.col.size();
  }
  return L;
}

void DistSpace::Exchange(Vector &local, hipStream_t s) const {
  if (halo_) halo_->Prolongate(local.Data(), s);
}

DistAmgSolver::DistAmgSolver(const Context &ctx, const amg::HostCsr &A, const std::vector<int> &off, const AmgOptions &opt)
    : ctx_(&ctx), opt_(opt) {
  const int size = ctx.comm ? ctx.comm->Size() : 1, rank = ctx.comm ? ctx.comm->Rank() : 0;
  PA_REQUIRE(A.nrows == A.ncols && (int)off.size() == size + 1 && off.front() == 0 && off.back() == A.nrows,
             "DistAmgSolver: the global matrix and the ranks' row ranges");
  height = width = off[(size_t)rank + 1] - off[(size_t)rank];
  std::vector<std::vector<int>> loff;
  const amg::Hierarchy h = amg::SetupBlocks(A, off, loff, opt.max_levels, opt.coarse_size, opt.theta);
  const size_t nl = h.A.size();
  const amg::HostCsr &Ac = h.A.back();
  const bool direct = Ac.nrows <= std::max(opt.coarse_size, 1) * 4 && Ac.nrows <= 2000;
  std::vector<amg::HostCsr> Rg(nl > 0 ? nl - 1 : 0);
  for (size_t l = 0; l + 1 < nl; l++) Rg[l] = amg::Transpose(h.P[l]);
  lv_.resize(nl);
  // the spaces: what every rank reads of every level (the last level of a direct solve is summed, not exchanged)
  for (size_t l = 0; l < nl; l++) {
    Level &L = lv_[l];
    L.space = std::make_unique<DistSpace>(rank, size, loff[l]);
    const bool last = l + 1 == nl;
    if (!(last && direct)) L.space->Need(h.A[l], loff[l]);
    if (!last) L.space->Need(Rg[l], loff[l + 1]);
    if (l > 0) L.space->Need(h.P[l - 1], loff[l - 1]);
    L.space->Finalize(ctx.comm);  // (collective, level by level on every rank)
    rows_.push_back(h.A[l].nrows), nnz_.push_back((int)h.A[l].nnz());
  }
  for (size_t l = 0; l < nl; l++) {
    Level &L = lv_[l];
    const bool last = l + 1 == nl;
    const int n_own = L.space->NumOwned(), n_loc = L.space->NumLocal();
    L.x.SetSize(std::max(n_loc, 1)), L.r.SetSize(std::max(n_loc, 1)), L.d.SetSize(std::max(n_loc, 1));
    L.b.SetSize(std::max(n_own, 1)), L.t.SetSize(std::max(n_own, 1));
    linalg::Fill(ctx, L.x, 0.0), linalg::Fill(ctx, L.r, 0.0), linalg::Fill(ctx, L.d, 0.0);
    if (!(last && direct)) {
      const amg::HostCsr Al = L.space->Localize(h.A[l], loff[l]);
      L.A = std::make_unique<DeviceCsr>(ctx, Al, false);
      std::vector<double> dv = l1_inverse(Al);
      if (dv.empty()) dv.push_back(0.0);
      L.dinv = upload(ctx, dv);
    }
    if (!last) {
      L.R = std::make_unique<DeviceCsr>(ctx, L.space->Localize(Rg[l], loff[l + 1]), false);
      L.P = std::make_unique<DeviceCsr>(ctx, lv_[l + 1].space->Localize(h.P[l], loff[l]), false);
    }
  }
  if (direct) {
    // my rows of the pseudo-inverse, all columns
    const int n = Ac.nrows, r0 = loff.back()[(size_t)rank], r1 = loff.back()[(size_t)rank + 1];
    std::vector<double> M((size_t)n * n, 0.0);
    for (int r = 0; r < n; r++)
      for (int a = Ac.rowptr[(size_t)r]; a < Ac.rowptr[(size_t)r + 1]; a++) M[(size_t)r * n + Ac.col[(size_t)a]] = Ac.val[(size_t)a];
    const std::vector<double> inv = pseudo_inverse(std::move(M), n);
    amg::HostCsr C;
    C.nrows = r1 - r0, C.ncols = n;
    C.rowptr.resize((size_t)C.nrows + 1);
    C.col.resize((size_t)C.nrows * n), C.val.resize((size_t)C.nrows * n);
    for (int r = 0; r <= C.nrows; r++) C.rowptr[(size_t)r] = r * n;
    for (int r = 0; r < C.nrows; r++)
      for (int c = 0; c < n; c++) C.col[(size_t)r * n + c] = c, C.val[(size_t)r * n + c] = inv[(size_t)(r0 + r) * n + c];
    Cinv_ = std::make_unique<DeviceCsr>(ctx, C, false);
    gb_.SetSize(std::max(n, 1));
  }
  for (Level &L : lv_) L.space->Release();
}

void DistAmgSolver::Smooth(const Level &L, bool zero_guess) const {
  // cheb4 above with the ghosts of what A multiplies filled first; x, r, d: local vectors, b, t: owned
  const Context &c = *ctx_;
  const int n = L.space->NumOwned();
  Vector xo(L.x.Data(), n), ro(L.r.Data(), n), dn(L.d.Data(), n), bo(L.b.Data(), n), to(L.t.Data(), n);
  const Vector dv(const_cast<double *>(L.dinv.Data()), n);
  if (zero_guess) {
    if (n) linalg::Copy(c, bo, ro), linalg::Fill(c, xo, 0.0);
  } else {
    L.space->Exchange(L.x, c.stream);
    if (n) L.A->Op().Mult(L.x, ro), linalg::AXPBY(c, 1.0, bo, -1.0, ro);
  }
  if (n) linalg::ChebyOrder0(c, 4.0 / 3.0, dv, ro, dn);
  for (int k = 1; k < opt_.smooth_order; k++) {
    const double sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / (2.0 * k + 3.0);
    L.space->Exchange(L.d, c.stream);
    if (n) L.A->Op().Mult(L.d, to), linalg::ChebyStep(c, sd, sr, dv, to, ro, dn, xo);
  }
  if (n) linalg::AXPY(c, 1.0, dn, xo);
}

void DistAmgSolver::Cycle(size_t l) const {
  const Context &c = *ctx_;
  const Level &L = lv_[l];
  const int n = L.space->NumOwned();
  Vector xo(L.x.Data(), n), ro(L.r.Data(), n), bo(L.b.Data(), n);
  if (l + 1 == lv_.size()) {
    if (Cinv_) {  // the right-hand side of every rank into one global vector, my rows of the inverse
      const int ng = L.space->NumGlobal();
      Vector gb(gb_.Data(), ng);
      if (c.comm && c.comm->Size() > 1) {
        linalg::Fill(c, gb, 0.0);
        if (n) PA_HIP(hipMemcpyAsync(gb.Data() + L.space->Offset(), bo.Data(), sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, c.stream));
        c.comm->AllReduceSum(gb.Data(), ng, c.stream);
        if (n) Cinv_->Op().Mult(gb, xo);
      } else if (n) {
        Cinv_->Op().Mult(bo, xo);
      }
    } else {
      Smooth(L, true);
      for (int it = 0; it < 3; it++) Smooth(L, false);
    }
    return;
  }
  Smooth(L, true);
  L.space->Exchange(L.x, c.stream);
  if (n) L.A->Op().Mult(L.x, ro), linalg::AXPBY(c, 1.0, bo, -1.0, ro);
  const Level &N = lv_[l + 1];
  const int nc = N.space->NumOwned();
  L.space->Exchange(L.r, c.stream);
  if (nc) {
    Vector nb(N.b.Data(), nc);
    L.R->Op().Mult(L.r, nb);
  }
  Cycle(l + 1);
  N.space->Exchange(N.x, c.stream);
  if (n) L.P->Op().AddMult(N.x, xo, 1.0);
  Smooth(L, false);
}

void DistAmgSolver::Mult(const Vector &b, Vector &x) const {
  PA_REQUIRE(b.Size() == height && x.Size() == height, "size mismatch in DistAmgSolver");
  const Context &c = *ctx_;
  const Level &L = lv_[0];
  if (height) {
    Vector bo(L.b.Data(), height);
    linalg::Copy(c, b, bo);
  }
  Cycle(0);
  if (height) {
    Vector xo(L.x.Data(), height);
    linalg::Copy(c, xo, x);
  }
}

// ---- AMS, distributed --------------------------------------------------------------------------------------------------------
DistAmsSolver::DistAmsSolver(const Context &ctx, const amg::HostCsr &A, const amg::HostCsr &G, const double *coords, int dim,
                             const std::vector<char> &ess_flag, const std::vector<int> &eoff, const std::vector<int> &voff,
                             const AmsOptions &opt)
    : ctx_(&ctx), opt_(opt) {
  const int size = ctx.comm ? ctx.comm->Size() : 1, rank = ctx.comm ? ctx.comm->Rank() : 0;
  PA_REQUIRE((int)eoff.size() == size + 1 && (int)voff.size() == size + 1 && eoff.back() == A.nrows && voff.back() == G.ncols,
             "DistAmsSolver: the ranks' edge / vertex ranges");
  height = width = eoff[(size_t)rank + 1] - eoff[(size_t)rank];
  const AmsHost h = ams_host(A, G, coords, dim, ess_flag, opt.singular, voff);
  const amg::HostCsr Gt = opt.singular ? amg::HostCsr() : amg::Transpose(h.Gb), Pit = amg::Transpose(h.Pi);
  // what every rank reads of the three spaces, then (collective, the same order on every rank) their exchange plans
  E_ = std::make_unique<DistSpace>(rank, size, eoff);
  V_ = std::make_unique<DistSpace>(rank, size, voff);
  W_ = std::make_unique<DistSpace>(rank, size, h.woff);
  E_->Need(A, eoff);
  E_->Need(Pit, h.woff);
  W_->Need(h.Pi, eoff);
  if (!opt.singular) E_->Need(Gt, voff), V_->Need(h.Gb, eoff);
  E_->Finalize(ctx.comm), V_->Finalize(ctx.comm), W_->Finalize(ctx.comm);
  const amg::HostCsr Al = E_->Localize(A, eoff);
  A_ = std::make_unique<DeviceCsr>(ctx, Al, false);
  std::vector<double> dv = l1_inverse(Al);
  if (dv.empty()) dv.push_back(0.0);
  dinv_ = upload(ctx, dv);
  Pi_ = std::make_unique<DeviceCsr>(ctx, W_->Localize(h.Pi, eoff), false);
  Pit_ = std::make_unique<DeviceCsr>(ctx, E_->Localize(Pit, h.woff), false);
  if (!opt.singular) {
    G_ = std::make_unique<DeviceCsr>(ctx, V_->Localize(h.Gb, eoff), false);
    Gt_ = std::make_unique<DeviceCsr>(ctx, E_->Localize(Gt, voff), false);
  }
  const int ne = std::max(E_->NumLocal(), 1);
  x_.SetSize(ne), r_.SetSize(ne), d_.SetSize(ne), t_.SetSize(std::max(height, 1));
  xg_.SetSize(std::max(V_->NumLocal(), 1)), bg_.SetSize(std::max(V_->NumOwned(), 1));
  xp_.SetSize(std::max(W_->NumLocal(), 1)), bp_.SetSize(std::max(W_->NumOwned(), 1));
  for (Vector *v : {&x_, &r_, &d_, &xg_, &xp_}) linalg::Fill(ctx, *v, 0.0);
  E_->Release(), V_->Release(), W_->Release();
  // the auxiliary problems (their levels' plans after the three above, in this order on every rank)
  if (!opt.singular) BG_ = std::make_unique<DistAmgSolver>(ctx, h.BG, voff, opt.amg);
  BPi_ = std::make_unique<DistAmgSolver>(ctx, h.BPi, h.woff, opt.amg);
}

void DistAmsSolver::Smooth(const Vector &b, bool zero_guess) const {
  const Context &c = *ctx_;
  const int n = height;
  Vector xo(x_.Data(), n), ro(r_.Data(), n), dn(d_.Data(), n), to(t_.Data(), n);
  const Vector dv(const_cast<double *>(dinv_.Data()), n);
  if (zero_guess) {
    if (n) linalg::Copy(c, b, ro), linalg::Fill(c, xo, 0.0);
  } else {
    E_->Exchange(x_, c.stream);
    if (n) A_->Op().Mult(x_, ro), linalg::AXPBY(c, 1.0, b, -1.0, ro);
  }
  if (n) linalg::ChebyOrder0(c, 4.0 / 3.0, dv, ro, dn);
  for (int k = 1; k < opt_.smooth_order; k++) {
    const double sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / (2.0 * k + 3.0);
    E_->Exchange(d_, c.stream);
    if (n) A_->Op().Mult(d_, to), linalg::ChebyStep(c, sd, sr, dv, to, ro, dn, xo);
  }
  if (n) linalg::AXPY(c, 1.0, dn, xo);
}

// x += T B T^T (b - A x): bc owned, xc local in the auxiliary space C
void DistAmsSolver::Correct(const DeviceCsr &T, const DeviceCsr &Tt, const DistAmgSolver &B, const DistSpace &C, const Vector &b,
                            Vector &bc, Vector &xc) const {
  const Context &c = *ctx_;
  const int n = height, nc = C.NumOwned();
  Vector xo(x_.Data(), n), ro(r_.Data(), n), bco(bc.Data(), nc), xco(xc.Data(), nc);
  E_->Exchange(x_, c.stream);
  if (n) A_->Op().Mult(x_, ro), linalg::AXPBY(c, 1.0, b, -1.0, ro);
  E_->Exchange(r_, c.stream);
  if (nc) Tt.Op().Mult(r_, bco);
  B.Mult(bco, xco);
  C.Exchange(xc, c.stream);
  if (n) T.Op().AddMult(xc, xo, 1.0);
}

void DistAmsSolver::Mult(const Vector &b, Vector &x) const {
  PA_REQUIRE(b.Size() == height && x.Size() == height, "size mismatch in DistAmsSolver");
  const Context &c = *ctx_;
  Vector xo(x_.Data(), height);
  if (initial_guess && height) linalg::Copy(c, x, xo);
  for (int it = 0; it < opt_.cycle_it; it++) {
    Smooth(b, it == 0 && !initial_guess);
    if (BG_) Correct(*G_, *Gt_, *BG_, *V_, b, bg_, xg_);
    Correct(*Pi_, *Pit_, *BPi_, *W_, b, bp_, xp_);
    if (BG_) Correct(*G_, *Gt_, *BG_, *V_, b, bg_, xg_);
    Smooth(b, false);
  }
  if (height) linalg::Copy(c, xo, x);
}

}  // namespace palace
