// Full assembly of a partially assembled operator into CSR on the device.
//
// Replaces CeedOperatorFullAssemble (reference fem/libceed/operator.cpp:455-523; called from
// BilinearForm::FullAssemble and ParOperator::ParallelAssemble, linalg/rap.cpp:84-152, when a coarse
// solver needs a matrix).  The sparsity pattern is the union of the element connectivities; the
// values are obtained from the operator's own apply by probing: the columns are coloured so that no
// two columns of one colour share a row (distance-2 colouring), one apply per colour recovers all
// entries of those columns.  Every entry is therefore exactly what Mult produces, for every element
// type, restriction kind and D-stage variant, with no second implementation of the element matrices.
// Meant for the coarsest levels (p = 1: ~33 colours on hexahedra); the cost is #colours applies.
#include <algorithm>
#include <numeric>

#include "pa_internal.hpp"


namespace pa {

namespace {

__global__ void k_probe_vector(const int n, const int32_t *__restrict__ color, const int c, double *__restrict__ x) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < n) x[d] = (color[d] == c) ? 1.0 : 0.0;
}

__global__ void k_extract(const long long nnz, const int32_t *__restrict__ row_of, const int32_t *__restrict__ col,
                          const int32_t *__restrict__ color, const int c, const double *__restrict__ y,
                          double *__restrict__ val) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nnz && color[col[k]] == c) val[k] = y[row_of[k]];
}

}  // namespace

void apply_for_assembly(pa_op *op, const double *x, double *y, hipStream_t s);

}  // namespace pa

using namespace pa;

namespace {
// temporaries and the result of pa_op_full_assemble are released if anything in it throws
struct AssembleGuard {
  pa_csr *m = nullptr;
  std::vector<void *> tmp;
  ~AssembleGuard() {
    for (void *p : tmp) (void)hipFree(p);
    if (m) pa_csr_destroy(m);
  }
  template <typename T>
  T *keep(T *p) {
    tmp.push_back(p);
    return p;
  }
  void release(void *p) {
    for (auto &q : tmp)
      if (q == p) q = nullptr;
  }
};

// Two-space operators (BilinearForm(trial, test)::FullAssemble, e.g. Atn of models/modeeigensolver.cpp:45-56): rows are test
// dofs, columns trial dofs.  Same probing as below; two columns may share a colour when no row holds both, which takes the
// transposed pattern (column -> rows) next to the pattern itself.
void assemble_two_space(pa_op *op, bool skip_zeros, hipStream_t s, pa_csr **out) {
  const int nr = op->height, nc = op->width;
  std::vector<int64_t> cnt((size_t)nr + 1, 0);
  for (const MixedSub *ms : op->msubs) {
    // (the gradient form has space_dim rows of test dofs per element: h_off of its test side lists them all)
    PA_REQUIRE(!ms->error && ms->ne > 0 && (int)ms->s1.h_off.size() == ms->ne * ms->s1.P && !ms->s2.h_off.empty() &&
                   ms->s2.h_off.size() % ((size_t)ms->ne * ms->s2.P) == 0,
               "sub-operator has no assembled form");
    for (int32_t r : ms->s2.h_off) cnt[(size_t)r + 1] += ms->s1.P;
  }
  for (int r = 0; r < nr; r++) cnt[r + 1] += cnt[r];
  PA_REQUIRE(cnt[nr] < (int64_t)1 << 31, "operator too large for full assembly (int32 CSR)");
  std::vector<int32_t> cols((size_t)cnt[nr]);
  {
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (const MixedSub *ms : op->msubs)
      for (int e = 0; e < ms->ne; e++) {
        const int rows_e = (int)(ms->s2.h_off.size() / (size_t)ms->ne);
        const int32_t *re = &ms->s2.h_off[(size_t)e * rows_e], *ce = &ms->s1.h_off[(size_t)e * ms->s1.P];
        for (int i = 0; i < rows_e; i++) {
          int64_t &f = fill[re[i]];
          for (int j = 0; j < ms->s1.P; j++) cols[(size_t)f++] = ce[j];
        }
      }
  }
  std::vector<int32_t> rowptr((size_t)nr + 1, 0), col;
  for (int r = 0; r < nr; r++) {
    auto b = cols.begin() + cnt[r], e = cols.begin() + cnt[r + 1];
    std::sort(b, e);
    e = std::unique(b, e);
    col.insert(col.end(), b, e);
    rowptr[r + 1] = (int32_t)col.size();
  }
  cols.clear();
  cols.shrink_to_fit();
  const int64_t nnz = (int64_t)col.size();
  // transposed pattern
  std::vector<int32_t> cptr((size_t)nc + 1, 0), rows_of_col((size_t)nnz), row_of((size_t)nnz);
  for (int64_t k = 0; k < nnz; k++) cptr[(size_t)col[k] + 1]++;
  for (int d = 0; d < nc; d++) cptr[d + 1] += cptr[d];
  {
    std::vector<int32_t> fill(cptr.begin(), cptr.end() - 1);
    for (int r = 0; r < nr; r++)
      for (int32_t a = rowptr[r]; a < rowptr[r + 1]; a++) rows_of_col[fill[col[a]]++] = r, row_of[a] = r;
  }
  // greedy colouring of the columns: d conflicts with every column of every row it appears in
  std::vector<int32_t> color((size_t)nc, -1), mark;
  int ncolors = 0;
  for (int d = 0; d < nc; d++) {
    if ((int)mark.size() < ncolors + 1) mark.resize(ncolors + 1, -1);
    for (int32_t a = cptr[d]; a < cptr[d + 1]; a++) {
      const int r = rows_of_col[a];
      for (int32_t b2 = rowptr[r]; b2 < rowptr[r + 1]; b2++) {
        const int cc = color[col[b2]];
        if (cc >= 0) mark[cc] = d;
      }
    }
    int c = 0;
    while (c < ncolors && mark[c] == d) c++;
    if (c == ncolors) ncolors++, mark.push_back(-1);
    color[d] = c;
  }
  AssembleGuard guard;
  auto *m = guard.m = new pa_csr;
  m->nrows = nr, m->ncols = nc;
  m->symmetric = false;
  int32_t *d_color = guard.keep(dev_upload(color.data(), color.size(), s)), *d_row_of = guard.keep(dev_upload(row_of.data(), row_of.size(), s));
  int32_t *d_col = guard.keep(dev_upload(col.data(), col.size(), s));
  double *d_val = guard.keep(dev_alloc<double>((size_t)std::max<int64_t>(nnz, 1))), *d_x = guard.keep(dev_alloc<double>((size_t)nc)),
         *d_y = guard.keep(dev_alloc<double>((size_t)nr));
  PA_HIP(hipMemsetAsync(d_val, 0, sizeof(double) * (size_t)nnz, s));
  for (int c = 0; c < ncolors; c++) {
    hipLaunchKernelGGL(k_probe_vector, dim3((nc + 255) / 256), dim3(256), 0, s, nc, d_color, c, d_x);
    apply_for_assembly(op, d_x, d_y, s);
    hipLaunchKernelGGL(k_extract, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, (long long)nnz, d_row_of, d_col, d_color, c,
                       d_y, d_val);
  }
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  if (skip_zeros) {  // operator.cpp:262-313: drop the entries that are exactly zero
    std::vector<double> val((size_t)nnz);
    PA_HIP(hipMemcpy(val.data(), d_val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
    std::vector<int32_t> rp((size_t)nr + 1, 0), cl;
    std::vector<double> vl;
    for (int r = 0; r < nr; r++) {
      for (int32_t a = rowptr[r]; a < rowptr[r + 1]; a++)
        if (val[a] != 0.0) cl.push_back(col[a]), vl.push_back(val[a]);
      rp[r + 1] = (int32_t)cl.size();
    }
    m->nnz = (int64_t)cl.size();
    m->d_rowptr = dev_upload(rp.data(), rp.size(), s);
    m->d_col = cl.empty() ? nullptr : dev_upload(cl.data(), cl.size(), s);
    m->d_val = vl.empty() ? nullptr : dev_upload(vl.data(), vl.size(), s);
  } else {
    m->nnz = nnz;
    m->d_rowptr = dev_upload(rowptr.data(), rowptr.size(), s);
    m->d_col = d_col, m->d_val = d_val;
    guard.release(d_col), guard.release(d_val);
  }
  guard.m = nullptr;
  *out = m;
}
}  // namespace

extern "C" {

int pa_op_full_assemble(pa_op *op, int skip_zeros, void *stream, pa_csr **out) {
  return guarded([&] {
    PA_REQUIRE(op && out, "null argument");
    PA_REQUIRE(op->finalized, "full assembly needs a finalized operator");
    hipStream_t s = (hipStream_t)stream;
    if (!op->msubs.empty()) {  // two-space operator: rectangular
      PA_REQUIRE(op->subs.empty() && op->dsubs.empty(), "two-space sub-operators are assembled on their own");
      assemble_two_space(op, skip_zeros != 0, s, out);
      return;
    }
    PA_REQUIRE(op->height == op->width, "full assembly needs a square operator");
    const int n = op->height;
    // ---- pattern: union of the element connectivities
    struct Conn {
      int ne, P;
      std::vector<int32_t> off;
    };
    std::vector<Conn> conns;
    for (const SubOp *so : op->subs) {
      Conn c{so->ne, so->P, {}};
      c.off.resize(so->h_sidx.size());
      for (size_t k = 0; k < c.off.size(); k++) c.off[k] = so->h_sidx[k] >= 0 ? so->h_sidx[k] : -1 - so->h_sidx[k];
      conns.push_back(std::move(c));
    }
    for (const DenseSub *ds : op->dsubs) conns.push_back(Conn{ds->ne, ds->P, ds->h_off});
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (const Conn &c : conns)
      for (size_t k = 0; k < c.off.size(); k++) cnt[(size_t)c.off[k] + 1] += c.P;
    for (int r = 0; r < n; r++) cnt[r + 1] += cnt[r];
    PA_REQUIRE(cnt[n] < (int64_t)1 << 31, "operator too large for full assembly (int32 CSR)");
    std::vector<int32_t> cols((size_t)cnt[n]);
    {
      std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
      for (const Conn &c : conns)
        for (int e = 0; e < c.ne; e++) {
          const int32_t *oe = &c.off[(size_t)e * c.P];
          for (int i = 0; i < c.P; i++) {
            int64_t &f = fill[oe[i]];
            for (int j = 0; j < c.P; j++) cols[(size_t)f++] = oe[j];
          }
        }
    }
    std::vector<int32_t> rowptr((size_t)n + 1, 0), col;
    col.reserve(cols.size() / 2);
    for (int r = 0; r < n; r++) {
      auto b = cols.begin() + cnt[r], e = cols.begin() + cnt[r + 1];
      std::sort(b, e);
      e = std::unique(b, e);
      col.insert(col.end(), b, e);
      rowptr[r + 1] = (int32_t)col.size();
    }
    cols.clear();
    cols.shrink_to_fit();
    const int64_t nnz = (int64_t)col.size();
    // ---- distance-2 colouring of the columns (symmetric pattern: neighbours of neighbours)
    std::vector<int32_t> color((size_t)n, -1), mark;
    int ncolors = 0;
    for (int d = 0; d < n; d++) {
      if ((int)mark.size() < ncolors + 1) mark.resize(ncolors + 1, -1);
      for (int32_t a = rowptr[d]; a < rowptr[d + 1]; a++) {
        const int r = col[a];
        for (int32_t b2 = rowptr[r]; b2 < rowptr[r + 1]; b2++) {
          const int cc = color[col[b2]];
          if (cc >= 0) mark[cc] = d;
        }
      }
      int c = 0;
      while (c < ncolors && mark[c] == d) c++;
      if (c == ncolors) ncolors++, mark.push_back(-1);
      color[d] = c;
    }
    // ---- values by probing
    std::vector<int32_t> row_of((size_t)nnz);
    for (int r = 0; r < n; r++)
      for (int32_t a = rowptr[r]; a < rowptr[r + 1]; a++) row_of[a] = r;
    AssembleGuard guard;
    auto *m = guard.m = new pa_csr;
    m->nrows = n;
    m->symmetric = op->symmetric();
    int32_t *d_color = guard.keep(dev_upload(color.data(), color.size(), s)), *d_row_of = guard.keep(dev_upload(row_of.data(), row_of.size(), s));
    int32_t *d_col = guard.keep(dev_upload(col.data(), col.size(), s));
    double *d_val = guard.keep(dev_alloc<double>((size_t)nnz)), *d_x = guard.keep(dev_alloc<double>((size_t)n)), *d_y = guard.keep(dev_alloc<double>((size_t)n));
    PA_HIP(hipMemsetAsync(d_val, 0, sizeof(double) * (size_t)nnz, s));
    for (int c = 0; c < ncolors; c++) {
      hipLaunchKernelGGL(k_probe_vector, dim3((n + 255) / 256), dim3(256), 0, s, n, d_color, c, d_x);
      apply_for_assembly(op, d_x, d_y, s);
      hipLaunchKernelGGL(k_extract, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, s, (long long)nnz, d_row_of, d_col,
                         d_color, c, d_y, d_val);
    }
    PA_HIP(hipGetLastError());
    PA_HIP(hipStreamSynchronize(s));
    if (skip_zeros) {  // operator.cpp:262-313: drop the entries that are exactly zero
      std::vector<double> val((size_t)nnz);
      PA_HIP(hipMemcpy(val.data(), d_val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
      std::vector<int32_t> rp((size_t)n + 1, 0), cl;
      std::vector<double> vl;
      cl.reserve((size_t)nnz), vl.reserve((size_t)nnz);
      for (int r = 0; r < n; r++) {
        for (int32_t a = rowptr[r]; a < rowptr[r + 1]; a++)
          if (val[a] != 0.0 || col[a] == r)  // the diagonal slot stays: EliminateEssential writes the DIAG_ONE value there
            cl.push_back(col[a]), vl.push_back(val[a]);
        rp[r + 1] = (int32_t)cl.size();
      }
      m->nnz = (int64_t)cl.size();
      m->d_rowptr = dev_upload(rp.data(), rp.size(), s);
      m->d_col = dev_upload(cl.data(), cl.size(), s);
      m->d_val = dev_upload(vl.data(), vl.size(), s);
    } else {
      m->nnz = nnz;
      m->d_rowptr = dev_upload(rowptr.data(), rowptr.size(), s);
      m->d_col = d_col, m->d_val = d_val;
      guard.release(d_col), guard.release(d_val);
    }
    guard.m = nullptr;
    *out = m;
  });
}

int pa_csr_get(const pa_csr *m, int32_t *nrows, int64_t *nnz, const int32_t **rowptr, const int32_t **colidx,
               const double **values) {
  return guarded([&] {
    PA_REQUIRE(m, "null argument");
    if (nrows) *nrows = m->nrows;  // (columns: pa_csr_num_cols)
    if (nnz) *nnz = m->nnz;
    if (rowptr) *rowptr = m->d_rowptr;
    if (colidx) *colidx = m->d_col;
    if (values) *values = m->d_val;
  });
}

int pa_csr_num_cols(const pa_csr *m) { return m ? (m->ncols ? m->ncols : m->nrows) : -1; }

void pa_csr_destroy(pa_csr *m) {
  if (!m) return;
  hipFree(m->d_rowptr), hipFree(m->d_col), hipFree(m->d_val);
  delete m;
}

}  // extern "C"
