// Device side of the native coarse-level solvers (see amg_solver.hpp): V-cycles of sparse matrix-vector products and fused
// vector kernels on the context's stream.
#include "amg_solver.hpp"

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
           Vector &r, Vector &d, Vector &t) {
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
}

void AmgSolver::Smooth(const Level &L, const Vector &b, Vector &x, bool zero_guess) const {
  cheb4(*ctx_, L.A->Op(), L.dinv, opt_.smooth_order, b, x, zero_guess, L.r, L.d, L.t);
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
  L.A->Op().Mult(x, L.r);
  linalg::AXPBY(c, 1.0, b, -1.0, L.r);
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

AmsSolver::AmsSolver(const Context &ctx, const amg::HostCsr &A, const amg::HostCsr &G, const double *coords, int dim,
                     const std::vector<char> &ess_flag, const AmsOptions &opt)
    : ctx_(&ctx), opt_(opt) {
  PA_REQUIRE(A.nrows == A.ncols && G.nrows == A.nrows, "AmsSolver: A [edges x edges] and G [edges x vertices] expected");
  PA_REQUIRE(dim >= 2 && dim <= 3 && coords, "AmsSolver needs the vertex coordinates");
  PA_REQUIRE((int)ess_flag.size() == A.nrows, "essential flag size mismatch");
  height = width = A.nrows;
  const int ne = A.nrows, nv = G.ncols;
  // transfers without the essential edges: no correction ever touches those
  const amg::HostCsr Gb = amg::DropRows(G, ess_flag);
  // Pi_c = |G| diag(G x_c) / 2, columns c nv + v (HYPRE_AMSSetCoordinateVectors; ams.cpp:64-100)
  amg::HostCsr Pi;
  Pi.nrows = ne, Pi.ncols = dim * nv;
  Pi.rowptr.assign((size_t)ne + 1, 0);
  std::vector<amg::HostCsr> Pic((size_t)dim);
  for (int c = 0; c < dim; c++) Pic[c].nrows = ne, Pic[c].ncols = nv, Pic[c].rowptr.assign((size_t)ne + 1, 0);
  for (int e = 0; e < ne; e++) {
    if (!ess_flag[e]) {
      for (int c = 0; c < dim; c++) {
        double tc = 0.0;
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) tc += G.val[a] * coords[(size_t)G.col[a] * dim + c];
        for (int a = G.rowptr[e]; a < G.rowptr[e + 1]; a++) {
          const double v = 0.5 * std::abs(G.val[a]) * tc;
          Pi.col.push_back(c * nv + G.col[a]), Pi.val.push_back(v);
          Pic[c].col.push_back(G.col[a]), Pic[c].val.push_back(v);
        }
      }
    }
    Pi.rowptr[e + 1] = (int)Pi.col.size();
    for (int c = 0; c < dim; c++) Pic[c].rowptr[e + 1] = (int)Pic[c].col.size();
  }
  // auxiliary matrices: G^T A G and the block-diagonal matrix of the Pi_c^T A Pi_c (one hierarchy serves the three additive
  // scalar corrections: its aggregates never cross the blocks)
  if (!opt.singular) BG_ = std::make_unique<AmgSolver>(ctx, galerkin(A, Gb), opt.amg);
  {
    amg::HostCsr B;
    B.nrows = B.ncols = dim * nv;
    B.rowptr.assign(1, 0);
    for (int c = 0; c < dim; c++) {
      const amg::HostCsr M = galerkin(A, Pic[c]);
      for (int r = 0; r < nv; r++) {
        for (int a = M.rowptr[r]; a < M.rowptr[r + 1]; a++) B.col.push_back(c * nv + M.col[a]), B.val.push_back(M.val[a]);
        B.rowptr.push_back((int)B.col.size());
      }
    }
    BPi_ = std::make_unique<AmgSolver>(ctx, B, opt.amg);
  }
  A_ = std::make_unique<DeviceCsr>(ctx, A, true);
  if (!opt.singular) {
    G_ = std::make_unique<DeviceCsr>(ctx, Gb, false);
    Gt_ = std::make_unique<DeviceCsr>(ctx, amg::Transpose(Gb), false);
    bg_.SetSize(nv), xg_.SetSize(nv);
  }
  Pi_ = std::make_unique<DeviceCsr>(ctx, Pi, false);
  Pit_ = std::make_unique<DeviceCsr>(ctx, amg::Transpose(Pi), false);
  bp_.SetSize(dim * nv), xp_.SetSize(dim * nv);
  dinv_ = upload(ctx, l1_inverse(A));
  r_.SetSize(ne), d_.SetSize(ne), t_.SetSize(ne);
}

void AmsSolver::Smooth(const Vector &b, Vector &x, bool zero_guess) const {
  cheb4(*ctx_, A_->Op(), dinv_, opt_.smooth_order, b, x, zero_guess, r_, d_, t_);
}

// x += T B T^T (b - A x)
void AmsSolver::Correct(const DeviceCsr &T, const DeviceCsr &Tt, const AmgSolver &B, const Vector &b, Vector &x, Vector &bc,
                        Vector &xc) const {
  A_->Op().Mult(x, r_);
  linalg::AXPBY(*ctx_, 1.0, b, -1.0, r_);
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

}  // namespace palace
