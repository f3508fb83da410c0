// Assembled (CSR) operator on the device: the coarsest multigrid level as a matrix.
//
// Reference: the coarsest level of the hierarchy is fully assembled (ParOperator::ParallelAssemble,
// linalg/rap.cpp:84-152, from CeedOperatorFullAssemble, fem/libceed/operator.cpp:455-523) because its solver
// (AMS / BoomerAMG through HYPRE) needs a matrix.  Here the matrix comes from pa_op_full_assemble and stays in
// HBM; the apply is a sparse matrix-vector product, which at p = 1 moves ~3x fewer bytes than the matrix-free
// operator carrying the fine level's quadrature data.
#include <cstdlib>
#include "linalg.hpp"

namespace palace {

namespace {

// one row per group of LPR lanes: coalesced (col, val) reads, the gather of x is the only indirection
template <int LPR>
__global__ __launch_bounds__(256) void k_csr_spmv(const int n, const int32_t *__restrict__ rowptr,
                                                  const int32_t *__restrict__ col, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *__restrict__ y, const double a,
                                                  const int add) {
  const int row = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPR);
  const int l = threadIdx.x % LPR;
  double s = 0.0;
  if (row < n) {
    const int32_t b = rowptr[row], e = rowptr[row + 1];
#ifdef PA_CSR_OLD  // (A / B builds: the loop of rounds 1-5)
    for (int32_t k = b + l; k < e; k += LPR) s += val[k] * x[col[k]];
#else
    // (round 6: the first four entries of the lane requested side by side -- col, val, then x -- instead of one dependent chain per
    // entry; same entries in the same order: same bits)
    int32_t c[4];
    double v[4], xv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int32_t k = b + l + q * LPR;
      c[q] = k < e ? col[k] : -1;
      v[q] = k < e ? val[k] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) xv[q] = c[q] >= 0 ? x[c[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (c[q] >= 0) s += v[q] * xv[q];
    for (int32_t k = b + l + 4 * LPR; k < e; k += LPR) s += val[k] * x[col[k]];
#endif
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LPR);
  if (row < n && l == 0) y[row] = add ? y[row] + a * s : a * s;
}

// y = A x consumed where it is produced (round 6: Operator::MultChebyStep / MultResidual for assembled matrices -- the smoothers of the
// algebraic levels are six launches of a few microseconds each per smoothing; with the step in the product's epilogue they are two):
//   mode 1:  out (+)= x + sd (x - ep) + sr dinv .* (r0 - A x)        (one step of the accumulated Chebyshev recurrence)
//   mode 2:  res = r0 - A x  and / or  out = sr dinv .* (r0 - A x)    (residual, first direction)
// `out`, `res` must not alias x (other rows read it).
struct CsrStep {
  int mode, add;
  double sd, sr;
  const double *dinv, *r0, *ep;
  double *out, *res;
};
template <int LPR>
__global__ __launch_bounds__(256) void k_csr_spmv_step(const int n, const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ x, const CsrStep st) {
  const int row = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPR);
  const int l = threadIdx.x % LPR;
  double s = 0.0;
  if (row < n) {
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    int32_t c[4];
    double v[4], xv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int32_t k = b + l + q * LPR;
      c[q] = k < e ? col[k] : -1;
      v[q] = k < e ? val[k] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) xv[q] = c[q] >= 0 ? x[c[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (c[q] >= 0) s += v[q] * xv[q];
    for (int32_t k = b + l + 4 * LPR; k < e; k += LPR) s += val[k] * x[col[k]];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LPR);
  if (row >= n || l != 0) return;
  const double rv = st.r0[row] - s;
  if (st.mode == 2) {
    if (st.res) st.res[row] = rv;
    if (st.out) st.out[row] = st.sr * st.dinv[row] * rv;
    return;
  }
  const double ev = x[row];
  double dk = st.sr * st.dinv[row] * rv;
  dk += st.sd * (ev - (st.ep ? st.ep[row] : 0.0));
  st.out[row] = (st.add ? st.out[row] : 0.0) + (ev + dk);
}

// the same on split vectors (multi-rank applies without L-vector copies): rows / columns [0, nsplit) live in y / x,
// the ghosts behind them in yg / one of two ghost buffers (parity of the device-resident counter *sel); xg*, yg shifted by -nsplit
template <int LPR>
__global__ __launch_bounds__(256) void k_csr_spmv_split(const int n, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ x, const double *__restrict__ xg0,
                                                        const double *__restrict__ xg1, const unsigned long long *__restrict__ sel,
                                                        double *__restrict__ y, double *__restrict__ yg, const int nsplit) {
  const int row = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) / LPR);
  const int l = threadIdx.x % LPR;
  const double *xg = ((sel ? *sel : 0ull) & 1ull) ? xg1 : xg0;
  double s = 0.0;
  if (row < n) {
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    for (int32_t k = b + l; k < e; k += LPR) {
      const int c = col[k];
      s += val[k] * (c < nsplit ? x : xg)[c];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LPR);
  if (row < n && l == 0) (row < nsplit ? y : yg)[row] = s;
}

// ParOperator's essential-dof handling applied to the matrix once (what the reference's ParallelAssemble does through
// EliminateBC, linalg/rap.cpp:131-149): rows and columns of essential dofs are zeroed, their diagonal is 1 or 0
__global__ void k_csr_eliminate(const int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                const double *__restrict__ val, const uint8_t *__restrict__ ess, const double diag,
                                double *__restrict__ out) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  const bool re = ess[row];
  for (int32_t k = rowptr[row]; k < rowptr[row + 1]; k++) {
    const int c = col[k];
    out[k] = (re || ess[c]) ? ((re && c == row) ? diag : 0.0) : val[k];
  }
}
__global__ void k_flag(const int n, const int32_t *__restrict__ list, uint8_t *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[list[i]] = 1;
}

__global__ void k_csr_diag(const int n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                           const double *__restrict__ val, double *__restrict__ d) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double s = 0.0;
  for (int32_t k = rowptr[row]; k < rowptr[row + 1]; k++)
    if (col[k] == row) s += val[k];
  d[row] = s;
}

}  // namespace

CsrOperator::CsrOperator(const Context &ctx, const pa_csr *m)
    : Operator(m->nrows, m->ncols ? m->ncols : m->nrows), ctx_(&ctx), m_(m) {
  const double avg = m->nrows ? (double)m->nnz / m->nrows : 0.0;
  lanes_ = avg > 24.0 ? 16 : (avg > 12.0 ? 8 : 4);
}

CsrOperator::~CsrOperator() {
}

double *CsrOperator::EliminatedValues(const int32_t *d_ess, int n_ess, bool diag_one) const {
  if (!height) return nullptr;
  double *d_val_bc_ = pa::dev_alloc<double>((size_t)m_->nnz);
  uint8_t *flag = pa::dev_alloc<uint8_t>((size_t)height);
  PA_HIP(hipMemsetAsync(flag, 0, (size_t)height, ctx_->stream));
  if (n_ess) hipLaunchKernelGGL(k_flag, dim3((n_ess + 255) / 256), dim3(256), 0, ctx_->stream, n_ess, d_ess, flag);
  hipLaunchKernelGGL(k_csr_eliminate, dim3((height + 255) / 256), dim3(256), 0, ctx_->stream, height, m_->d_rowptr,
                     m_->d_col, m_->d_val, flag, diag_one ? 1.0 : 0.0, d_val_bc_);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(ctx_->stream));
  PA_HIP(hipFree(flag));
  return d_val_bc_;
}

void CsrOperator::MultValues(const double *d_vals, const Vector &x, Vector &y) const { Apply(d_vals, x, y, 1.0, false); }

void CsrOperator::MultSplit(const double *d_vals, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel,
                            double *y, double *yg, int n_true) const {
  PA_REQUIRE(height == width && n_true >= 0 && n_true <= height, "split apply of a square assembled operator");
  if (!height) return;
  const double *vals = d_vals ? d_vals : m_->d_val;
  const long long threads = (long long)height * lanes_;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  const double *g0 = xg0 - n_true, *g1 = (xg1 ? xg1 : xg0) - n_true;
  double *go = yg - n_true;
#define PA_CSR_SPLIT(L)                                                                                                   \
  hipLaunchKernelGGL(k_csr_spmv_split<L>, grid, block, 0, ctx_->stream, height, m_->d_rowptr, m_->d_col, vals, x, g0, g1, sel, y, go, \
                     n_true)
  if (lanes_ == 16)
    PA_CSR_SPLIT(16);
  else if (lanes_ == 8)
    PA_CSR_SPLIT(8);
  else
    PA_CSR_SPLIT(4);
#undef PA_CSR_SPLIT
  PA_HIP(hipGetLastError());
}

void CsrOperator::Apply(const double *vals, const Vector &x, Vector &y, double a, bool add) const {
  PA_REQUIRE(x.Size() == width && y.Size() == height, "size mismatch in CsrOperator");
  if (!height) return;
  const long long threads = (long long)height * lanes_;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (lanes_ == 16)
    hipLaunchKernelGGL(k_csr_spmv<16>, grid, block, 0, ctx_->stream, height, m_->d_rowptr, m_->d_col, vals, x.Data(),
                       y.Data(), a, (int)add);
  else if (lanes_ == 8)
    hipLaunchKernelGGL(k_csr_spmv<8>, grid, block, 0, ctx_->stream, height, m_->d_rowptr, m_->d_col, vals, x.Data(),
                       y.Data(), a, (int)add);
  else
    hipLaunchKernelGGL(k_csr_spmv<4>, grid, block, 0, ctx_->stream, height, m_->d_rowptr, m_->d_col, vals, x.Data(),
                       y.Data(), a, (int)add);
  PA_HIP(hipGetLastError());
}

namespace {
void launch_csr_step(const Context &c, const pa_csr *m, int lanes, int n, const double *x, const CsrStep &st, const double *vals = nullptr) {
  if (!n) return;
  if (!vals) vals = m->d_val;
  const long long threads = (long long)n * lanes;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  if (lanes == 16) hipLaunchKernelGGL(k_csr_spmv_step<16>, grid, block, 0, c.stream, n, m->d_rowptr, m->d_col, vals, x, st);
  else if (lanes == 8) hipLaunchKernelGGL(k_csr_spmv_step<8>, grid, block, 0, c.stream, n, m->d_rowptr, m->d_col, vals, x, st);
  else hipLaunchKernelGGL(k_csr_spmv_step<4>, grid, block, 0, c.stream, n, m->d_rowptr, m->d_col, vals, x, st);
  PA_HIP(hipGetLastError());
}
}  // namespace
bool CsrOperator::PrepareChebyStep() const {
  const char *e = std::getenv("PALACE_AMD_FUSED_STEP_CSR");  // (read at every set-up: A / B runs in one process)
  return height == width && !(e && e[0] == '0');
}
void CsrOperator::MultChebyStep(const Vector &x, const ChebyStepArgs &a) const {
  PA_REQUIRE(x.Size() == width && a.out && a.out->Size() == height && a.dinv && a.r0, "size mismatch in CsrOperator::MultChebyStep");
  PA_REQUIRE(a.out->Data() != x.Data(), "CsrOperator::MultChebyStep: the result must not alias the vector the matrix multiplies");
  launch_csr_step(*ctx_, m_, lanes_, height, x.Data(),
                  CsrStep{1, a.add ? 1 : 0, a.sd, a.sr, a.dinv->Data(), a.r0->Data(), a.e_prev ? a.e_prev->Data() : nullptr, a.out->Data(), nullptr});
}
void CsrOperator::MultChebyStepValues(const double *d_vals, const Vector &x, const ChebyStepArgs &a) const {
  PA_REQUIRE(x.Size() == width && a.out && a.out->Size() == height && a.dinv && a.r0, "size mismatch in CsrOperator::MultChebyStep");
  PA_REQUIRE(a.out->Data() != x.Data(), "CsrOperator::MultChebyStep: the result must not alias the vector the matrix multiplies");
  launch_csr_step(*ctx_, m_, lanes_, height, x.Data(),
                  CsrStep{1, a.add ? 1 : 0, a.sd, a.sr, a.dinv->Data(), a.r0->Data(), a.e_prev ? a.e_prev->Data() : nullptr, a.out->Data(), nullptr},
                  d_vals);
}
void CsrOperator::MultResidualValues(const double *d_vals, const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0,
                                     Vector *d0) const {
  PA_REQUIRE(y.Size() == width && b.Size() == height && (res || d0) && (!d0 || dinv), "bad arguments of CsrOperator::MultResidual");
  PA_REQUIRE((!res || res->Data() != y.Data()) && (!d0 || d0->Data() != y.Data()), "CsrOperator::MultResidual: results must not alias y");
  launch_csr_step(*ctx_, m_, lanes_, height, y.Data(),
                  CsrStep{2, 0, 0.0, c0, dinv ? dinv->Data() : nullptr, b.Data(), nullptr, d0 ? d0->Data() : nullptr, res ? res->Data() : nullptr},
                  d_vals);
}
void CsrOperator::MultResidual(const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0, Vector *d0) const {
  PA_REQUIRE(y.Size() == width && b.Size() == height && (res || d0) && (!d0 || dinv), "bad arguments of CsrOperator::MultResidual");
  PA_REQUIRE((!res || res->Data() != y.Data()) && (!d0 || d0->Data() != y.Data()), "CsrOperator::MultResidual: results must not alias y");
  launch_csr_step(*ctx_, m_, lanes_, height, y.Data(),
                  CsrStep{2, 0, 0.0, c0, dinv ? dinv->Data() : nullptr, b.Data(), nullptr, d0 ? d0->Data() : nullptr, res ? res->Data() : nullptr});
}
void CsrOperator::Mult(const Vector &x, Vector &y) const { Apply(m_->d_val, x, y, 1.0, false); }
void CsrOperator::MultTranspose(const Vector &x, Vector &y) const {
  PA_REQUIRE(m_->symmetric, "MultTranspose of an assembled non-symmetric operator is not available");
  Apply(m_->d_val, x, y, 1.0, false);
}
void CsrOperator::AddMult(const Vector &x, Vector &y, double a) const { Apply(m_->d_val, x, y, a, true); }
void CsrOperator::AssembleDiagonal(Vector &diag) const {
  PA_REQUIRE(diag.Size() == height, "size mismatch in CsrOperator::AssembleDiagonal");
  if (!height) return;
  hipLaunchKernelGGL(k_csr_diag, dim3((height + 255) / 256), dim3(256), 0, ctx_->stream, height, m_->d_rowptr, m_->d_col,
                     m_->d_val, diag.Data());
  PA_HIP(hipGetLastError());
}

}  // namespace palace
