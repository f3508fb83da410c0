// C ABI of libpalace_amd.so (declared in include/palace_amd.h): object lifetime, descriptor
// validation and set-up on the host; all arithmetic lives in the HIP kernels.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "pa_internal.hpp"

namespace pa {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

static void require_device() {
  int n = 0;
  const hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess || n == 0)
    throw Error(std::string("no HIP device visible: libpalace_amd has no CPU fallback (hipGetDeviceCount: ") +
                hipGetErrorString(err) + ", " + std::to_string(n) + " devices)");
}

// coeff_qf.h:7-45 — [nattr][attr->mat ...][nmat][nmat*dim*dim doubles]; 8-byte slots.
void parse_coeff(const void *blob, size_t bytes, int dim, CoeffHost &out, size_t slot_offset) {
  const size_t nslots = bytes / 8;
  const unsigned char *base = static_cast<const unsigned char *>(blob);
  auto slot_int = [&](size_t i) {
    PA_REQUIRE(slot_offset + i < nslots, "coefficient context truncated");
    int32_t v;
    std::memcpy(&v, base + 8 * (slot_offset + i), 4);
    return v;
  };
  auto slot_dbl = [&](size_t i) {
    PA_REQUIRE(slot_offset + i < nslots, "coefficient context truncated");
    double v;
    std::memcpy(&v, base + 8 * (slot_offset + i), 8);
    return v;
  };
  const int nattr = slot_int(0);
  PA_REQUIRE(nattr >= 0, "negative attribute count in coefficient context");
  out.attr_mat.resize(nattr);
  for (int i = 0; i < nattr; i++) out.attr_mat[i] = slot_int(1 + i);
  const int nmat = slot_int(1 + nattr);
  PA_REQUIRE(nmat > 0, "coefficient context without materials");
  out.dim = dim;
  out.mat.resize((size_t)nmat * dim * dim);
  for (size_t i = 0; i < out.mat.size(); i++) out.mat[i] = slot_dbl(2 + nattr + i);
  for (int i = 0; i < nattr; i++)
    PA_REQUIRE(out.attr_mat[i] >= 0 && out.attr_mat[i] < nmat, "attribute maps to missing material");
  out.slots = 2 + nattr + out.mat.size();
  out.d_attr_mat = nattr ? dev_upload(out.attr_mat.data(), (size_t)nattr) : nullptr;
  out.d_mat = dev_upload(out.mat.data(), out.mat.size());
  // transposed copy for A^T (only when it differs)
  std::vector<double> mt(out.mat);
  bool sym = true;
  for (int k = 0; k < nmat; k++)
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) {
        mt[(size_t)k * dim * dim + i + dim * j] = out.mat[(size_t)k * dim * dim + j + dim * i];
        sym = sym && out.mat[(size_t)k * dim * dim + i + dim * j] == out.mat[(size_t)k * dim * dim + j + dim * i];
      }
  out.d_mat_t = sym ? nullptr : dev_upload(mt.data(), mt.size());
}

static thread_local bool g_transpose = false;
TransposeScope::TransposeScope(bool on) : prev_(g_transpose) { g_transpose = on; }
TransposeScope::~TransposeScope() { g_transpose = prev_; }
bool TransposeScope::active() { return g_transpose; }
CoeffDev CoeffHost::dev() const {
  return CoeffDev{d_attr_mat, (g_transpose && d_mat_t) ? d_mat_t : d_mat, (int)attr_mat.size()};
}

static int expected_P(int fe_type, int p) {
  return fe_type == PA_FE_HCURL ? 3 * p * (p + 1) * (p + 1) : (p + 1) * (p + 1) * (p + 1);
}

// Dense table of the tensor element (native dof order) from the 1-D tables, to validate what the
// caller passed as DofToQuad::FULL data (basis.cpp:43-83).
static void check_dense_tables(const pa_basis_desc &b, int P, int Q) {
  if (!b.interp && !b.deriv) return;
  const int p = b.order, q1 = b.q1d, nc = p + 1;
  auto nat = [&](int l, double &sgn) {
    int n = b.dof_map ? b.dof_map[l] : l;
    sgn = 1.0;
    if (n < 0) n = -1 - n, sgn = -1.0;
    return n;
  };
  double worst = 0.0;
  if (b.fe_type == PA_FE_HCURL) {
    for (int C = 0; C < 3; C++) {
      const int ni = C == 0 ? p : nc, nj = C == 1 ? p : nc, nk = C == 2 ? p : nc;
      const double *TX = C == 0 ? b.Bo : b.Bc, *TY = C == 1 ? b.Bo : b.Bc, *TZ = C == 2 ? b.Bo : b.Bc;
      for (int k = 0; k < nk; k++)
        for (int j = 0; j < nj; j++)
          for (int i = 0; i < ni; i++) {
            double sgn;
            const int n = nat(C * p * nc * nc + i + ni * (j + nj * k), sgn);
            for (int qz = 0; qz < q1; qz++)
              for (int qy = 0; qy < q1; qy++)
                for (int qx = 0; qx < q1; qx++) {
                  const int q = qx + q1 * (qy + q1 * qz);
                  const double bx = TX[qx * ni + i], by = TY[qy * nj + j], bz = TZ[qz * nk + k];
                  const double gx = C == 0 ? 0 : b.Gc[qx * nc + i], gy = C == 1 ? 0 : b.Gc[qy * nc + j],
                               gz = C == 2 ? 0 : b.Gc[qz * nc + k];
                  const double f = sgn * bx * by * bz;
                  const double dx = sgn * gx * by * bz, dy = sgn * bx * gy * bz, dz = sgn * bx * by * gz;
                  double val[3] = {0, 0, 0}, cv[3];
                  val[C] = f;
                  if (C == 0) cv[0] = 0, cv[1] = dz, cv[2] = -dy;
                  if (C == 1) cv[0] = -dz, cv[1] = 0, cv[2] = dx;
                  if (C == 2) cv[0] = dy, cv[1] = -dx, cv[2] = 0;
                  for (int d = 0; d < 3; d++) {
                    if (b.interp) worst = std::fmax(worst, std::fabs(b.interp[((size_t)d * Q + q) * P + n] - val[d]));
                    if (b.deriv) worst = std::fmax(worst, std::fabs(b.deriv[((size_t)d * Q + q) * P + n] - cv[d]));
                  }
                }
          }
    }
  } else {
    for (int k = 0; k < nc; k++)
      for (int j = 0; j < nc; j++)
        for (int i = 0; i < nc; i++) {
          double sgn;
          const int n = nat(i + nc * (j + nc * k), sgn);
          for (int qz = 0; qz < q1; qz++)
            for (int qy = 0; qy < q1; qy++)
              for (int qx = 0; qx < q1; qx++) {
                const int q = qx + q1 * (qy + q1 * qz);
                const double bx = b.Bc[qx * nc + i], by = b.Bc[qy * nc + j], bz = b.Bc[qz * nc + k];
                const double gx = b.Gc[qx * nc + i], gy = b.Gc[qy * nc + j], gz = b.Gc[qz * nc + k];
                if (b.interp) worst = std::fmax(worst, std::fabs(b.interp[(size_t)q * P + n] - bx * by * bz));
                if (b.deriv) {
                  const double gr[3] = {gx * by * bz, bx * gy * bz, bx * by * gz};
                  for (int d = 0; d < 3; d++)
                    worst = std::fmax(worst, std::fabs(b.deriv[((size_t)d * Q + q) * P + n] - gr[d]));
                }
              }
        }
  }
  if (!(worst < 1e-10))
    throw Error("dense basis table does not match the tensor product of the 1-D tables (max diff " +
                std::to_string(worst) + ")");
}

static SubOp *make_sub(pa_geom *geom, const pa_restriction_desc &r, const pa_basis_desc &b, int qf,
                       const void *ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops,
                       QData *shared_qd = nullptr) {
  require_device();
  PA_REQUIRE(geom && geom->d_geom, "geometry data missing");
  PA_REQUIRE(geom->eb == 0, "geometry data of a dense element block: use pa_op_add_sub_dense");
  PA_REQUIRE(!r.curl_orients, "the curl-oriented restriction needs the dense-table path (pa_op_add_sub_dense)");
  PA_REQUIRE(b.fe_type == PA_FE_HCURL || b.fe_type == PA_FE_H1, "unknown finite element type");
  PA_REQUIRE(b.order >= 1 && b.order + 1 <= kMaxP1 + 1, "unsupported element order");
  PA_REQUIRE(b.q1d == geom->q1d, "basis and geometry data use different quadrature rules");
  PA_REQUIRE(b.Bc && b.Gc && (b.fe_type == PA_FE_H1 || b.Bo), "1-D basis tables missing");
  PA_REQUIRE(r.num_elem == geom->ne, "restriction and geometry data disagree on element count");
  const int P = expected_P(b.fe_type, b.order);
  PA_REQUIRE(r.elem_size == P, "restriction element size does not match the basis");
  PA_REQUIRE(r.offsets, "restriction offsets missing");
  const bool cross = qf == PA_QF_HCURLHDIV_33 || qf == PA_QF_HDIVHCURL_33;
  if (cross) {  // integ/mixedveccurl.cpp:21-120 on one H(curl) space
    PA_REQUIRE(b.fe_type == PA_FE_HCURL, "the mixed curl QFunctions need an H(curl) space");
    const uint32_t ti = qf == PA_QF_HCURLHDIV_33 ? PA_EVAL_INTERP : PA_EVAL_CURL;
    const uint32_t te = qf == PA_QF_HCURLHDIV_33 ? PA_EVAL_CURL : PA_EVAL_INTERP;
    PA_REQUIRE(trial_ops == ti && test_ops == te, "evaluation modes do not match the QFunction's inputs");
  } else {
    PA_REQUIRE(trial_ops == test_ops, "trial and test evaluation modes differ only for the mixed curl QFunctions");
  }
  uint32_t want = 0;
  switch (qf) {
    case PA_QF_HCURLHDIV_33:
    case PA_QF_HDIVHCURL_33: want = trial_ops; break;
    case PA_QF_HDIV_33: want = PA_EVAL_CURL; break;
    case PA_QF_HCURL_33: want = b.fe_type == PA_FE_HCURL ? PA_EVAL_INTERP : PA_EVAL_GRAD; break;
    case PA_QF_HDIVMASS_33: want = PA_EVAL_CURL | PA_EVAL_INTERP; break;
    case PA_QF_HCURLMASS_33: want = PA_EVAL_GRAD | PA_EVAL_INTERP; break;
    case PA_QF_H1_1: want = PA_EVAL_INTERP; break;
    default: throw Error("unknown QFunction id");
  }
  PA_REQUIRE(trial_ops == want, "evaluation modes do not match the QFunction's inputs");
  const bool nd_qf = qf == PA_QF_HDIV_33 || qf == PA_QF_HDIVMASS_33 || cross ||
                     (qf == PA_QF_HCURL_33 && b.fe_type == PA_FE_HCURL);
  PA_REQUIRE(nd_qf == (b.fe_type == PA_FE_HCURL), "QFunction does not match the element type");

  auto *so = new SubOp;
  so->geom = geom;
  geom->refcount++;
  so->fe_type = b.fe_type, so->p = b.order, so->q1d = b.q1d, so->P = P, so->Q = geom->Q;
  so->ne = r.num_elem, so->lsize = r.lsize, so->qf = qf;
  so->trial_ops = trial_ops, so->test_ops = test_ops;
  const int nc = b.order + 1;
  so->Bc.assign(b.Bc, b.Bc + b.q1d * nc);
  so->Gc.assign(b.Gc, b.Gc + b.q1d * nc);
  if (b.Bo) so->Bo.assign(b.Bo, b.Bo + b.q1d * b.order);
  check_dense_tables(b, P, geom->Q);
  // the kernels rely on the mirror symmetry of Gauss-Legendre / Gauss-Lobatto tables
  {
    double worst = 0.0;
    const int q1 = b.q1d;
    for (int q = 0; q < q1; q++) {
      for (int i = 0; i < nc; i++) {
        worst = std::fmax(worst, std::fabs(so->Bc[q * nc + i] - so->Bc[(q1 - 1 - q) * nc + (nc - 1 - i)]));
        worst = std::fmax(worst, std::fabs(so->Gc[q * nc + i] + so->Gc[(q1 - 1 - q) * nc + (nc - 1 - i)]));
      }
      if (b.Bo)
        for (int i = 0; i < b.order; i++)
          worst = std::fmax(worst, std::fabs(so->Bo[q * b.order + i] -
                                             so->Bo[(q1 - 1 - q) * b.order + (b.order - 1 - i)]));
    }
    if (!(worst < 1e-12)) {
      delete so;
      geom->refcount--;
      throw Error("1-D basis tables are not mirror-symmetric (non Gauss-Legendre/Lobatto nodes?)");
    }
    std::vector<double> tab;
    tab.insert(tab.end(), so->Bo.begin(), so->Bo.end());
    if (so->Bo.empty()) tab.assign((size_t)b.q1d * b.order, 0.0);
    tab.insert(tab.end(), so->Bc.begin(), so->Bc.end());
    tab.insert(tab.end(), so->Gc.begin(), so->Gc.end());
    so->d_tab = dev_upload(tab.data(), tab.size());
  }

  // signed tensor-order index array (restriction.cpp:290-296 semantics folded with dof_map)
  std::vector<int32_t> lidx((size_t)r.num_elem * P);
  std::vector<char> seen(P);
  for (int l = 0; l < P; l++) {
    int n = b.dof_map ? b.dof_map[l] : l;
    if (n < 0) n = -1 - n;
    PA_REQUIRE(n >= 0 && n < P && !seen[n], "dof_map is not a signed permutation");
    seen[n] = 1;
  }
  for (int e = 0; e < r.num_elem; e++)  // internal element order (the geometry data's, pa_geom.hip)
    for (int l = 0; l < P; l++) {
      int n = b.dof_map ? b.dof_map[l] : l;
      bool neg = false;
      if (n < 0) n = -1 - n, neg = true;
      const size_t k = (size_t)(geom->eorder.empty() ? e : geom->eorder[e]) * P + n;
      const int32_t off = r.offsets[k];
      PA_REQUIRE(off >= 0 && off < r.lsize, "restriction offset out of range");
      if (r.orients && r.orients[k]) neg = !neg;
      lidx[(size_t)e * P + l] = neg ? -1 - off : off;
    }
  so->d_lidx = dev_upload(lidx.data(), lidx.size());
  PA_REQUIRE(r.lsize < kEssBit, "too many local dofs for the index encoding");
  PA_REQUIRE(P < 65536, "element too large for the 16-bit slot permutation");
  // Sorted order of the element's entries (by global dof, stable): what E gathers and E^T stores
  // in, so that one load/store instruction touches neighbouring dofs.
  const size_t nnz = lidx.size();
  std::vector<int32_t> sidx(nnz);
  std::vector<uint16_t> perm(nnz);
  {
    std::vector<int> ord(P);
    for (int e = 0; e < r.num_elem; e++) {
      const int32_t *le = &lidx[(size_t)e * P];
      for (int l = 0; l < P; l++) ord[l] = l;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int c2) {
        const int da = le[a] >= 0 ? le[a] : -1 - le[a], dc = le[c2] >= 0 ? le[c2] : -1 - le[c2];
        return da < dc;
      });
      for (int m = 0; m < P; m++) {
        sidx[(size_t)e * P + m] = le[ord[m]];
        perm[(size_t)e * P + m] = (uint16_t)ord[m];
      }
    }
  }
  so->d_sidx = dev_upload(sidx.data(), nnz);
  so->d_perm = dev_upload(perm.data(), nnz);
  so->h_perm = perm;
  // transpose map for the gather form of E^T (counting sort by dof; element order preserved, so the
  // summation order of every dof is fixed) unless PALACE_AMD_SCATTER=atomic asks for the atomic form
  const char *mode = getenv("PALACE_AMD_SCATTER");
  if (!(mode && std::string(mode) == "atomic") || b.fe_type == PA_FE_H1) {
    std::vector<int32_t> tptr((size_t)r.lsize + 1, 0), tent(nnz);
    for (size_t k = 0; k < nnz; k++) {
      const int32_t s = sidx[k];
      tptr[(size_t)(s >= 0 ? s : -1 - s) + 1]++;
    }
    for (int d = 0; d < r.lsize; d++) tptr[d + 1] += tptr[d];
    std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
    for (size_t k = 0; k < nnz; k++) {  // E-vector position of a sorted entry is its own index
      const int32_t s = sidx[k];
      const int d = s >= 0 ? s : -1 - s;
      tent[fill[d]++] = s >= 0 ? (int32_t)k : -1 - (int32_t)k;
    }
    so->d_tptr = dev_upload(tptr.data(), tptr.size());
    so->d_tent = dev_upload(tent.data(), tent.size());
    so->d_ye = dev_alloc<double>((size_t)((r.num_elem + 3) & ~3) * P);  // padded to whole batches of the streaming kernel
  }
  so->h_sidx = std::move(sidx);

  so->ctx_blob.assign((const uint8_t *)ctx, (const uint8_t *)ctx + ctx_size);
  PA_REQUIRE(ctx && ctx_size >= 24 && ctx_size % 8 == 0, "coefficient context missing or malformed");
  switch (qf) {
    case PA_QF_HDIV_33:
    case PA_QF_HCURL_33:
    case PA_QF_HCURLHDIV_33:
    case PA_QF_HDIVHCURL_33:
      parse_coeff(ctx, ctx_size, 3, so->c0, 0);
      break;
    case PA_QF_HDIVMASS_33:
      parse_coeff(ctx, ctx_size, 3, so->c0, 0);
      parse_coeff(ctx, ctx_size, 3, so->c1, so->c0.slots);
      break;
    case PA_QF_HCURLMASS_33:
      parse_coeff(ctx, ctx_size, 1, so->c0, 0);
      parse_coeff(ctx, ctx_size, 3, so->c1, so->c0.slots);
      break;
    case PA_QF_H1_1:
      parse_coeff(ctx, ctx_size, 1, so->c0, 0);
      break;
  }
  auto is_iso = [](const CoeffHost &c) {
    if (c.dim != 3) return true;
    for (size_t k = 0; k + 9 <= c.mat.size(); k += 9)
      for (int i = 0; i < 9; i++) {
        const bool diag = (i % 4 == 0);
        if (diag ? c.mat[k + i] != c.mat[k] : c.mat[k + i] != 0.0) return false;
      }
    return true;
  };
  so->iso = !cross && is_iso(so->c0) && (so->c1.mat.empty() || is_iso(so->c1));
  // Pre-assembled packed symmetric D (default for symmetric coefficients; PALACE_AMD_QDATA=0 keeps
  // the reference-default matrix-free D from the geometry factors)
  auto is_sym = [](const CoeffHost &c) {
    if (c.dim != 3) return true;
    for (size_t k = 0; k + 9 <= c.mat.size(); k += 9)
      if (c.mat[k + 1] != c.mat[k + 3] || c.mat[k + 2] != c.mat[k + 6] || c.mat[k + 5] != c.mat[k + 7]) return false;
    return true;
  };
  const char *qmode = getenv("PALACE_AMD_QDATA");
  const bool want_qd = !cross && !(qmode && std::string(qmode) == "0") && is_sym(so->c0) &&  // (mixed forms: matrix-free D)
                       (so->c1.mat.empty() || is_sym(so->c1));
  // D stage of the curl-curl + mass operator on H(curl) hexes when every coefficient is isotropic: the metric
  // form, 7 doubles per point shared by all such operators and p-levels of a mesh instead of 12 per operator
  // (measured: +2 % PCG iterations/s, half the q-data memory; for curl-curl or mass alone the packed D is 3-15 %
  // faster, so those keep it).  PALACE_AMD_DSTAGE=qdata / metric force one form for every H(curl) hex operator.
  const char *dmode = getenv("PALACE_AMD_DSTAGE");
  const std::string dm = dmode ? dmode : "";
  const bool metric_ok = b.fe_type == PA_FE_HCURL && want_qd && so->iso && dm != "qdata" &&
                         (qf == PA_QF_HDIVMASS_33 || dm == "metric") && geom->d_attr_e && (int)geom->w1.size() == b.q1d;
  if (metric_ok) {
    if (!geom->metric) launch_nd_hex_metric(*so, nullptr);
    so->qd = geom->metric;
    geom->metric->refcount++;
  }
  if (so->qd) {
    // metric form chosen above
  } else if (shared_qd) {
    so->qd = shared_qd;
    shared_qd->refcount++;
  } else if (want_qd) {
    if (b.fe_type == PA_FE_HCURL)
      launch_nd_hex_qdata(*so, nullptr);
    else
      launch_h1_hex_qdata(*so, nullptr);
    PA_HIP(hipStreamSynchronize(nullptr));
  }
  return so;
}

static void free_sub(SubOp *so) {
  if (!so) return;
  hipFree(so->d_lidx);
  hipFree(so->d_sidx), hipFree(so->d_sidx_bc), hipFree(so->d_perm), hipFree(so->d_perm_x), hipFree(so->d_shared), hipFree(so->d_shared_bc);
  hipFree(so->d_ye), hipFree(so->d_ye2), hipFree(so->d_tptr), hipFree(so->d_tent);
  free_stream(*so);
  if (so->qd && --so->qd->refcount == 0) {
    hipFree(so->qd->d), hipFree(so->qd->d_aff);
    delete so->qd;
  }
  hipFree(so->d_tab);
  hipFree(so->c0.d_attr_mat), hipFree(so->c0.d_mat), hipFree(so->c0.d_mat_t);
  hipFree(so->c1.d_attr_mat), hipFree(so->c1.d_mat), hipFree(so->c1.d_mat_t);
  pa_geom_destroy(static_cast<pa_geom *>(so->geom));
  delete so;
}

// y (+)= A x.  overwrite: the first sub-operator writes y instead of accumulating (Mult without a
// separate memset when E^T runs as a gather).
// after: an event the entries of x that take part in the halo exchange wait for (multi-rank applies, pa_op_mult_after): a
// streaming block with interface batch lists runs its interior batches first; everything else simply waits up front
// timing experiments (scripts/price_evec_cache.py): 1 = element kernel only, 2 = E^T run gather only (of the streaming form)
#ifdef PA_ABLATION  // (the ablation library only, `make ablate`: the product library has no such switch)
static int g_debug_phase = 0;
extern "C" void pa_debug_apply_phase(int phase) { g_debug_phase = phase; }
#else
constexpr int g_debug_phase = 0;
#endif

static void apply(pa_op *op, const double *x, double *y, bool overwrite, hipStream_t s, bool masked = false,
                  int ess_policy = -1, hipEvent_t after = nullptr) {
  PA_REQUIRE(op && x && y, "null argument");
  PA_REQUIRE(!op->subs.empty() || !op->dsubs.empty() || !op->msubs.empty(), "operator has no sub-operators");
  PA_REQUIRE(x != y, "in-place apply is not supported");
  PA_REQUIRE(op->msubs.empty() || !masked, "mixed-space sub-operators have no essential-dof form");
  PA_REQUIRE(op->msubs.empty() || !TransposeScope::active() || (op->subs.empty() && op->dsubs.empty()),
             "transposed apply of an operator with two-space sub-operators: those only");
  bool first = true;
  const bool split = after && op->subs.size() == 1 && op->dsubs.empty() && op->subs[0]->fe_type == PA_FE_HCURL &&
                     op->subs[0]->d_idxc && op->subs[0]->has_blist && overwrite && (!masked || op->subs[0]->d_perm_s_bc);
  if (after && !split) PA_HIP(hipStreamWaitEvent(s, after, 0));
  for (const SubOp *so : op->subs) {
    if (so->fe_type == PA_FE_HCURL) {
      if (split) {
        launch_nd_hex_stream(*so, x, y, masked, s, 0);  // batches that touch no exchanged dof
        PA_HIP(hipStreamWaitEvent(s, after, 0));
        launch_nd_hex_stream(*so, x, y, masked, s, 1);
        launch_et_run_gather(*so, y, false, s, x, masked, ess_policy);
      } else if (so->d_idxc && overwrite && first && (!masked || so->d_perm_s_bc)) {  // streaming kernel (y = A x) + E^T of the shared dofs by runs
        if (g_debug_phase != 2) launch_nd_hex_stream(*so, x, y, masked, s);
        if (g_debug_phase != 1) launch_et_run_gather(*so, y, false, s, x, masked, ess_policy);
      } else if (so->d_ye) {
        launch_nd_hex_apply(*so, x, y, so->d_ye, masked, s, !(overwrite && first), ess_policy);
        launch_et_gather(*so, y, !(overwrite && first), s, x, ess_policy);
      } else {
        if (overwrite && first) PA_HIP(hipMemsetAsync(y, 0, sizeof(double) * (size_t)op->height, s));
        launch_nd_hex_apply(*so, x, y, nullptr, masked, s);
      }
    } else if (so->d_idxc && so->stream_default && overwrite && first && (!masked || so->d_perm_s_bc)) {  // streaming form, see above
      launch_h1_hex_stream(*so, x, y, masked, s);
      launch_et_run_gather(*so, y, false, s, x, masked, ess_policy);
    } else {
      launch_h1_hex_apply(*so, x, masked, s);
      launch_et_gather(*so, y, !(overwrite && first), s);
    }
    first = false;
  }
  for (const DenseSub *ds : op->dsubs) {
    launch_dense_apply(*ds, x, masked, s);
    launch_dense_gather(*ds, y, !(overwrite && first), s);
    first = false;
  }
  for (const MixedSub *ms : op->msubs) {
    launch_mixed_apply(*ms, x, y, !(overwrite && first), s, TransposeScope::active());
    first = false;
  }
}

// Operators with a single H(curl) hex block: dofs with exactly one element copy skip the E-vector
// (PALACE_AMD_DIRECT=0 keeps every dof on the gather path).  Dense sub-operators next to it (round 5: the surface terms of a
// volume operator) do not change that: apply() runs the hex block first, overwriting, and they accumulate after its gather.
void finalize_exclusive(pa_op *op) {
  const char *mode = getenv("PALACE_AMD_DIRECT");
  if (mode && std::string(mode) == "0") return;
  if (op->subs.size() != 1 || !op->msubs.empty()) return;
  SubOp *so = op->subs[0];
  if (!so->d_ye || so->d_perm_x || !so->h_shared.empty()) return;
  if (so->fe_type == PA_FE_H1) {
    // H1 blocks: only the streaming kernel stores exclusive dofs directly; the one-shot kernel and its gather keep the
    // full dof list (no d_perm_x / d_shared), the streaming arrays get the list of shared dofs
    if (!h1_hex_stream_capable(*so)) return;
    so->stream_default = h1_hex_stream_ok(*so);
    std::vector<int32_t> cnt((size_t)so->lsize, 0);
    for (const int32_t s : so->h_sidx) cnt[s >= 0 ? s : -1 - s]++;
    for (int d = 0; d < so->lsize; d++)
      if (cnt[d] != 1) so->h_shared.push_back(d);
    so->n_shared = (int)so->h_shared.size();
    build_stream(*so);
    return;
  }
  if (so->fe_type != PA_FE_HCURL) return;
  const size_t nnz = so->h_sidx.size();
  std::vector<int32_t> count((size_t)so->lsize, 0);
  for (size_t k = 0; k < nnz; k++) count[so->h_sidx[k] >= 0 ? so->h_sidx[k] : -1 - so->h_sidx[k]]++;
  std::vector<uint16_t> px(so->h_perm);
  size_t nex = 0;
  for (size_t k = 0; k < nnz; k++)
    if (count[so->h_sidx[k] >= 0 ? so->h_sidx[k] : -1 - so->h_sidx[k]] == 1) px[k] |= (uint16_t)kExclBit16, nex++;
  if (nex == 0) return;
  std::vector<int32_t> shared;
  shared.reserve((size_t)so->lsize);
  // dofs without any copy (count == 0) stay on the list so that Mult still writes their zero
  for (int d = 0; d < so->lsize; d++)
    if (count[d] != 1) shared.push_back(d);
  so->d_perm_x = dev_upload(px.data(), px.size());
  so->d_shared = dev_upload(shared.data(), shared.size());
  so->n_shared = (int)shared.size();
  so->h_shared = std::move(shared);
  build_stream(*so);
}

// y0 = A x0, y1 = A x1 (overwrite).  One pass over the index / q-data streams when the operator is a single
// H(curl) hex block on the q-data path, else two applies.
static bool apply2(pa_op *op, const double *x0, const double *x1, double *y0, double *y1, hipStream_t s, bool masked,
                   int ess_policy) {
  PA_REQUIRE(op && x0 && x1 && y0 && y1, "null argument");
  PA_REQUIRE(x0 != y0 && x0 != y1 && x1 != y0 && x1 != y1 && y0 != y1, "in-place apply is not supported");
  if (op->subs.size() == 1 && op->dsubs.empty() && nd_hex_supports_two_rhs(*op->subs[0])) {
    SubOp *so = op->subs[0];
    if (!so->d_ye2) so->d_ye2 = dev_alloc<double>((size_t)((so->ne + 3) & ~3) * so->P);
    const int pol = nd_hex_fuses_essential(*so) ? ess_policy : -1;
    launch_nd_hex_apply(*so, x0, y0, so->d_ye, masked, s, false, pol, x1, y1, so->d_ye2);
    launch_et_gather2(*so, y0, y1, false, s, x0, x1, pol);
    return pol >= 0;
  }
  apply(op, x0, y0, true, s, masked, -1);
  apply(op, x1, y1, true, s, masked, -1);
  return false;
}

void apply_for_assembly(pa_op *op, const double *x, double *y, hipStream_t s) { apply(op, x, y, true, s); }

}  // namespace pa

static uint64_t next_op_id() {
  static std::atomic<uint64_t> counter{0};
  return ++counter;
}

bool pa_op::symmetric() const {
  for (const pa::SubOp *so : subs)  // (C u, curl v) and (C curl u, v) are each other's transposes, never their own
    if (so->qf == PA_QF_HCURLHDIV_33 || so->qf == PA_QF_HDIVHCURL_33) return false;
  for (const pa::SubOp *so : subs)
    if (!so->c0.symmetric() || !so->c1.symmetric()) return false;
  for (const pa::DenseSub *ds : dsubs)
    if (!ds->c0.symmetric() || !ds->c1.symmetric()) return false;
  if (!msubs.empty()) return false;
  return true;
}

using namespace pa;

extern "C" {

const char *pa_last_error(void) { return g_error.c_str(); }

const char *pa_version(void) { return "palace_amd 0.1 (gfx950)"; }

int pa_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int pa_geom_create(const pa_mesh_desc *mesh, void *stream, pa_geom **geom) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(mesh && geom, "null argument");
    PA_REQUIRE(mesh->num_elem > 0 && mesh->mesh_order >= 1 && mesh->q1d >= 1 && mesh->q1d <= kMaxQ1,
               "bad mesh descriptor");
    PA_REQUIRE(mesh->node_offsets && mesh->nodes && mesh->attr && mesh->mesh_B && mesh->mesh_G &&
                   mesh->qweight1d,
               "mesh descriptor arrays missing");
    auto *g = new pa_geom;
    try {
      launch_geom(*mesh, *g, (hipStream_t)stream);
    } catch (...) {
      delete g;
      throw;
    }
    *geom = g;
  });
}

int pa_geom_create_dense(const pa_mesh_dense_desc *mesh, void *stream, pa_geom **geom) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(mesh && geom, "null argument");
    auto *g = new pa_geom;
    try {
      launch_geom_dense(*mesh, *g, (hipStream_t)stream);
    } catch (...) {
      delete g;
      throw;
    }
    *geom = g;
  });
}

int pa_geom_layout(const pa_geom *geom, int32_t out[4]) {
  return guarded([&] {
    PA_REQUIRE(geom && out, "null argument");
    out[0] = geom->ne, out[1] = geom->Q, out[2] = geom->eb ? geom->Qpad : geom->Q, out[3] = geom->eb;
  });
}

int pa_geom_num_rows(const pa_geom *geom) { return geom ? geom->nrows : -1; }

int pa_geom_element_order(const pa_geom *geom, int32_t *order) {
  return guarded([&] {
    PA_REQUIRE(geom && order, "null argument");
    for (int e = 0; e < geom->ne; e++) order[e] = geom->eorder.empty() ? e : geom->eorder[e];
  });
}

int pa_geom_retain(pa_geom *geom) {
  return guarded([&] {
    PA_REQUIRE(geom, "null argument");
    geom->refcount++;
  });
}

void pa_geom_destroy(pa_geom *geom) {
  if (!geom) return;
  if (--geom->refcount == 0) {
    hipFree(geom->d_geom);
    hipFree(geom->d_qw);
    hipFree(geom->d_attr_e);
    hipFree(geom->d_xnodes), hipFree(geom->d_gtab);
    if (geom->metric) {
      hipFree(geom->metric->d), hipFree(geom->metric->d_aff);
      delete geom->metric;
    }
    delete geom;
  }
}

int pa_geom_data(const pa_geom *geom, const double **dev_ptr, size_t *count) {
  return guarded([&] {
    PA_REQUIRE(geom && dev_ptr && count, "null argument");
    *dev_ptr = geom->d_geom;
    *count = geom->eb ? (size_t)((geom->ne + geom->eb - 1) / geom->eb) * geom->nrows * geom->Qpad * geom->eb
                      : (size_t)geom->ne * 11 * geom->Q;
  });
}

int pa_op_create(int32_t height, int32_t width, pa_op **op) {
  return guarded([&] {
    PA_REQUIRE(op && height > 0 && width > 0, "bad operator size");
    auto *o = new pa_op;
    o->id = next_op_id();
    o->height = height, o->width = width;
    *op = o;
  });
}

int pa_op_add_sub(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                  const pa_basis_desc *basis, int32_t qfunction, const void *ctx, size_t ctx_size,
                  uint32_t trial_ops, uint32_t test_ops) {
  return guarded([&] {
    PA_REQUIRE(op && geom && restr && basis, "null argument");
    PA_REQUIRE(op->height == op->width, "only square operators are supported");
    PA_REQUIRE(restr->lsize == op->width, "dimensions mismatch for sub-operator");  // operator.cpp:69-71
    op->subs.push_back(make_sub(geom, *restr, *basis, qfunction, ctx, ctx_size, trial_ops, test_ops));
    op->finalized = false;
  });
}

int pa_op_add_sub_dense(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                        const pa_dense_basis_desc *basis, int32_t qfunction, const void *ctx,
                        size_t ctx_size, uint32_t trial_ops, uint32_t test_ops) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(op && geom && restr && basis, "null argument");
    PA_REQUIRE(!op->finalized, "operator already finalized");
    PA_REQUIRE(op->height == op->width, "only square operators are supported");
    op->dsubs.push_back(
        make_dense_sub(geom, *restr, *basis, qfunction, ctx, ctx_size, trial_ops, test_ops, op->height));
  });
}

int pa_op_add_sub_dense_mixed(pa_op *op, pa_geom *geom, const pa_restriction_desc *trial_restr,
                              const pa_dense_basis_desc *trial_basis, const pa_restriction_desc *test_restr,
                              const pa_dense_basis_desc *test_basis, int32_t qfunction, const void *ctx, size_t ctx_size) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(op && geom && trial_restr && trial_basis && test_restr && test_basis, "null argument");
    PA_REQUIRE(!op->finalized, "operator already finalized");
    switch (qfunction) {
      case PA_QF_HCURLHDIV_33: case PA_QF_HDIVHCURL_33: case PA_QF_HCURL_33:
      case PA_QF_HCURLHDIV_22: case PA_QF_HDIVHCURL_22: case PA_QF_HCURL_22:
      case PA_QF_HCURLHDIV_32: case PA_QF_HDIVHCURL_32: case PA_QF_HCURL_32:
      case PA_QF_HCURLHDIV_31: case PA_QF_HDIVHCURL_31: case PA_QF_HCURL_31:
      case PA_QF_HCURLHDIV_21: case PA_QF_HDIVHCURL_21: case PA_QF_HCURL_21:
      case PA_QF_H1_1: case PA_QF_HDIV_33: break;
      default: throw Error("not a mixed-space QFunction");
    }
    PA_REQUIRE(test_restr->lsize == op->height && trial_restr->lsize == op->width,
               "dimensions mismatch for sub-operator");  // operator.cpp:69-71
    op->msubs.push_back(make_mixed_sub(geom, *trial_restr, *trial_basis, *test_restr, *test_basis, qfunction, ctx, ctx_size));
  });
}

int pa_op_add_sub_dense_gradient(pa_op *op, pa_geom *geom, const pa_restriction_desc *trial_restr,
                                 const pa_dense_basis_desc *trial_basis, const pa_restriction_desc *test_restr,
                                 const pa_dense_basis_desc *test_basis, int32_t comp_stride, int32_t qfunction, const void *ctx,
                                 size_t ctx_size) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(op && geom && trial_restr && trial_basis && test_restr && test_basis, "null argument");
    PA_REQUIRE(!op->finalized, "operator already finalized");
    PA_REQUIRE(test_restr->lsize == op->height && trial_restr->lsize == op->width,
               "dimensions mismatch for sub-operator");  // operator.cpp:69-71
    op->msubs.push_back(make_mixed_gradient_sub(geom, *trial_restr, *trial_basis, *test_restr, *test_basis, comp_stride, qfunction,
                                                ctx, ctx_size));
  });
}

int pa_op_add_sub_dense_vector_mass(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr, const pa_dense_basis_desc *basis,
                                    int32_t num_comp, int32_t comp_stride, const void *ctx, size_t ctx_size) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(op && geom && restr && basis && ctx, "null argument");
    PA_REQUIRE(!op->finalized, "operator already finalized");
    PA_REQUIRE((num_comp == 2 || num_comp == 3) && comp_stride >= 1, "MassIntegrator: 1 (pa_op_add_sub_dense), 2 or 3 components");
    PA_REQUIRE(restr->lsize == op->height && restr->lsize == op->width, "dimensions mismatch for sub-operator");
    PA_REQUIRE(basis->fe_type == PA_FE_H1 && !restr->orients && !restr->curl_orients, "vector mass: a scalar H1 basis per component");
    // f_apply_h1_2 | _3: v_r = w detJ sum_c C[r][c] u_c -- one scalar mass (f_apply_h1_1 between two scalar spaces) per non-zero
    // entry of the coefficient, from component c to component r
    CoeffHost C;
    parse_coeff(ctx, ctx_size, num_comp, C, 0);
    const int nattr = (int)C.attr_mat.size(), nmat = (int)(C.mat.size() / ((size_t)num_comp * num_comp));
    hipFree(C.d_attr_mat), hipFree(C.d_mat), hipFree(C.d_mat_t);
    const size_t ndof = (size_t)restr->num_elem * restr->elem_size;
    std::vector<std::vector<int32_t>> off((size_t)num_comp, std::vector<int32_t>(ndof));
    for (int c = 0; c < num_comp; c++)
      for (size_t k = 0; k < ndof; k++) {
        const int64_t v = (int64_t)restr->offsets[k] + (int64_t)c * comp_stride;
        PA_REQUIRE(restr->offsets[k] >= 0 && v < restr->lsize, "component offset out of range (comp_stride and lsize of the vector space)");
        off[(size_t)c][k] = (int32_t)v;
      }
    auto slot_int = [](int32_t v) {
      double d = 0.0;
      std::memcpy(&d, &v, 4);
      return d;
    };
    for (int r = 0; r < num_comp; r++)
      for (int c = 0; c < num_comp; c++) {
        std::vector<double> blob;  // the 1 x 1 context of entry (r, c): same attributes, one scalar per material
        blob.push_back(slot_int(nattr));
        for (int a = 0; a < nattr; a++) blob.push_back(slot_int(C.attr_mat[(size_t)a]));
        blob.push_back(slot_int(nmat));
        bool any = false;
        for (int m = 0; m < nmat; m++) {
          const double v = C.mat[(size_t)m * num_comp * num_comp + r + (size_t)num_comp * c];
          any = any || v != 0.0;
          blob.push_back(v);
        }
        if (!any) continue;
        pa_restriction_desc rt = *restr, rs = *restr;
        rt.offsets = off[(size_t)c].data(), rs.offsets = off[(size_t)r].data();
        op->msubs.push_back(make_mixed_sub(geom, rt, *basis, rs, *basis, PA_QF_H1_1, blob.data(), blob.size() * sizeof(double)));
      }
  });
}

struct pa_error_op {
  MixedSub *ms = nullptr;
};

int pa_error_op_create(pa_geom *geom, const pa_restriction_desc *restr1, const pa_dense_basis_desc *basis1,
                       const pa_restriction_desc *restr2, const pa_dense_basis_desc *basis2, int32_t qfunction,
                       const void *ctx, size_t ctx_size, pa_error_op **out) {
  return guarded([&] {
    require_device();
    PA_REQUIRE(geom && restr1 && basis1 && restr2 && basis2 && out, "null argument");
    PA_REQUIRE(qfunction == PA_QF_HCURLHDIV_ERROR_33 || qfunction == PA_QF_HDIVHCURL_ERROR_33 ||
                   qfunction == PA_QF_HCURLHDIV_ERROR_22 || qfunction == PA_QF_HDIVHCURL_ERROR_22 ||
                   qfunction == PA_QF_L2H1_ERROR,
               "not an error QFunction");
    auto *e = new pa_error_op;
    try {
      e->ms = make_mixed_sub(geom, *restr1, *basis1, *restr2, *basis2, qfunction, ctx, ctx_size);
    } catch (...) {
      delete e;
      throw;
    }
    *out = e;
  });
}

int pa_error_op_apply_add(pa_error_op *e, const double *u1, const double *u2, double *estimates, void *stream) {
  return guarded([&] {
    PA_REQUIRE(e && e->ms && u1 && u2 && estimates, "null argument");
    launch_mixed_error(*e->ms, u1, u2, estimates, (hipStream_t)stream);
  });
}

int pa_error_op_num_elem(const pa_error_op *e) { return (e && e->ms) ? e->ms->ne : -1; }

void pa_error_op_destroy(pa_error_op *e) {
  if (!e) return;
  free_mixed_sub(e->ms);
  delete e;
}

// Weighted sum of H(curl) integrators on one (geometry, space) pair as ONE sub-operator: D is linear in the material
// coefficient, so sum_k a_k {K(mu^-1), M(eps), C(sigma), ...} is a single curl-curl + mass pass whose two contexts are
// the weighted sums of the terms' contexts (per attribute).  Returns the combined blob (mass first, like the
// reference's pair contexts) and the QFunction / eval modes of the fused integrator.
static std::vector<uint8_t> combine_hcurl_terms(int nterms, const int32_t *qfs, const void *const *ctxs,
                                                const size_t *ctx_sizes, const double *coeffs, int dim, int &qf_out,
                                                uint32_t &ops_out) {
  PA_REQUIRE(nterms >= 1 && qfs && ctxs && ctx_sizes && coeffs, "bad argument");
  struct Part {
    std::vector<int32_t> attr_mat;
    std::vector<double> mat;
    double a;
  };
  std::vector<Part> mass, curl;
  auto read = [&](const void *blob, size_t bytes, size_t slot_offset, Part &out) -> size_t {
    const size_t nslots = bytes / 8;
    const unsigned char *base = static_cast<const unsigned char *>(blob);
    auto slot_int = [&](size_t i) {
      PA_REQUIRE(slot_offset + i < nslots, "coefficient context truncated");
      int32_t v;
      std::memcpy(&v, base + 8 * (slot_offset + i), 4);
      return v;
    };
    const int nattr = slot_int(0);
    PA_REQUIRE(nattr >= 0, "negative attribute count in coefficient context");
    out.attr_mat.resize(nattr);
    for (int i = 0; i < nattr; i++) out.attr_mat[i] = slot_int(1 + i);
    const int nmat = slot_int(1 + nattr);
    PA_REQUIRE(nmat > 0, "coefficient context without materials");
    out.mat.resize((size_t)nmat * dim * dim);
    PA_REQUIRE(slot_offset + 2 + nattr + out.mat.size() <= nslots, "coefficient context truncated");
    std::memcpy(out.mat.data(), base + 8 * (slot_offset + 2 + nattr), 8 * out.mat.size());
    for (int i = 0; i < nattr; i++)
      PA_REQUIRE(out.attr_mat[i] >= 0 && out.attr_mat[i] < nmat, "attribute maps to missing material");
    return 2 + nattr + out.mat.size();
  };
  for (int k = 0; k < nterms; k++) {
    PA_REQUIRE(ctxs[k], "null context");
    if (qfs[k] == PA_QF_HCURL_33) {
      mass.emplace_back();
      mass.back().a = coeffs[k];
      read(ctxs[k], ctx_sizes[k], 0, mass.back());
    } else if (qfs[k] == PA_QF_HDIV_33) {
      curl.emplace_back();
      curl.back().a = coeffs[k];
      read(ctxs[k], ctx_sizes[k], 0, curl.back());
    } else if (qfs[k] == PA_QF_HDIVMASS_33) {
      mass.emplace_back(), curl.emplace_back();
      mass.back().a = curl.back().a = coeffs[k];
      const size_t used = read(ctxs[k], ctx_sizes[k], 0, mass.back());
      read(ctxs[k], ctx_sizes[k], used, curl.back());
    } else {
      throw Error("only the H(curl) curl-curl / mass / curl-curl+mass integrators can be fused into one sum");
    }
  }
  int nattr = 0;
  for (const auto *v : {&mass, &curl})
    for (const Part &t : *v) {
      if (t.attr_mat.empty()) continue;
      PA_REQUIRE(nattr == 0 || nattr == (int)t.attr_mat.size(), "terms disagree on the number of mesh attributes");
      nattr = (int)t.attr_mat.size();
    }
  const int nout = std::max(nattr, 1), dd = dim * dim;
  auto pack = [&](const std::vector<Part> &parts, std::vector<uint8_t> &blob) {
    std::vector<double> m((size_t)nout * dd, 0.0);
    for (const Part &t : parts)
      for (int i = 0; i < nout; i++) {
        const int src = t.attr_mat.empty() ? 0 : t.attr_mat[i];
        for (int j = 0; j < dd; j++) m[(size_t)i * dd + j] += t.a * t.mat[(size_t)src * dd + j];
      }
    const size_t slots = 2 + (size_t)nattr + m.size(), at = blob.size();
    blob.resize(at + 8 * slots, 0);
    auto put_int = [&](size_t i, int32_t v) { std::memcpy(blob.data() + at + 8 * i, &v, 4); };
    put_int(0, nattr);
    for (int i = 0; i < nattr; i++) put_int(1 + i, i);
    put_int(1 + nattr, nout);
    std::memcpy(blob.data() + at + 8 * (2 + (size_t)nattr), m.data(), 8 * m.size());
  };
  std::vector<uint8_t> blob;
  if (!mass.empty()) pack(mass, blob);
  if (!curl.empty()) pack(curl, blob);
  qf_out = (!mass.empty() && !curl.empty()) ? PA_QF_HDIVMASS_33 : (!mass.empty() ? PA_QF_HCURL_33 : PA_QF_HDIV_33);
  ops_out = (!mass.empty() ? (uint32_t)PA_EVAL_INTERP : 0u) | (!curl.empty() ? (uint32_t)PA_EVAL_CURL : 0u);
  return blob;
}

int pa_op_add_sub_sum(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr, const pa_basis_desc *basis,
                      int32_t nterms, const int32_t *qfunctions, const void *const *ctxs, const size_t *ctx_sizes,
                      const double *coeffs) {
  int rc = guarded([&] {
    PA_REQUIRE(op && geom && restr && basis, "null argument");
    PA_REQUIRE(basis->fe_type == PA_FE_HCURL, "fused sums are built for H(curl) spaces");
  });
  if (rc) return rc;
  int qf = 0;
  uint32_t ops = 0;
  std::vector<uint8_t> blob;
  rc = guarded([&] { blob = combine_hcurl_terms(nterms, qfunctions, ctxs, ctx_sizes, coeffs, 3, qf, ops); });
  if (rc) return rc;
  return pa_op_add_sub(op, geom, restr, basis, qf, blob.data(), blob.size(), ops, ops);
}

int pa_op_add_sub_dense_sum(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                            const pa_dense_basis_desc *basis, int32_t nterms, const int32_t *qfunctions,
                            const void *const *ctxs, const size_t *ctx_sizes, const double *coeffs) {
  int rc = guarded([&] {
    PA_REQUIRE(op && geom && restr && basis, "null argument");
    PA_REQUIRE(basis->fe_type == PA_FE_HCURL && geom->dim == 3 && geom->sdim == 3,
               "fused sums are built for 3-D H(curl) spaces");
  });
  if (rc) return rc;
  int qf = 0;
  uint32_t ops = 0;
  std::vector<uint8_t> blob;
  rc = guarded([&] { blob = combine_hcurl_terms(nterms, qfunctions, ctxs, ctx_sizes, coeffs, 3, qf, ops); });
  if (rc) return rc;
  return pa_op_add_sub_dense(op, geom, restr, basis, qf, blob.data(), blob.size(), ops, ops);
}

int pa_op_finalize(pa_op *op) {
  return guarded([&] {
    PA_REQUIRE(op, "null argument");
    PA_REQUIRE(!op->subs.empty() || !op->dsubs.empty() || !op->msubs.empty(), "operator has no sub-operators");
    finalize_exclusive(op);
    op->finalized = true;
  });
}

int pa_op_coarsen(const pa_op *fine, const pa_restriction_desc *restr, const pa_basis_desc *basis,
                  pa_op **coarse) {
  return guarded([&] {
    PA_REQUIRE(fine && restr && basis && coarse, "null argument");
    PA_REQUIRE(!fine->subs.empty(), "fine operator has no sub-operators");
    auto *o = new pa_op;
    o->id = next_op_id();
    o->height = o->width = restr->lsize;
    try {
      for (const SubOp *fs : fine->subs) {
        PA_REQUIRE(fs->ne == restr->num_elem, "coarsening needs one element block (same elements)");
        o->subs.push_back(make_sub(static_cast<pa_geom *>(fs->geom), *restr, *basis, fs->qf,
                                   fs->ctx_blob.data(), fs->ctx_blob.size(), fs->trial_ops,
                                   fs->test_ops, fs->qd));
      }
    } catch (...) {
      pa_op_destroy(o);
      throw;
    }
    finalize_exclusive(o);
    o->finalized = true;
    *coarse = o;
  });
}

int pa_op_coarsen_dense(const pa_op *fine, const pa_restriction_desc *restr, const pa_dense_basis_desc *basis,
                        pa_op **coarse) {
  return guarded([&] {
    PA_REQUIRE(fine && restr && basis && coarse, "null argument");
    PA_REQUIRE(fine->finalized && fine->subs.empty() && !fine->dsubs.empty(), "fine operator has no dense sub-operators");
    auto *o = new pa_op;
    o->id = next_op_id();
    o->height = o->width = restr->lsize;
    try {
      for (const DenseSub *fs : fine->dsubs)
        o->dsubs.push_back(make_dense_sub(static_cast<pa_geom *>(fs->geom), *restr, *basis, fs->qf, fs->ctx_blob.data(),
                                          fs->ctx_blob.size(), fs->trial_ops, fs->test_ops, o->height, fs->contra));
    } catch (...) {
      pa_op_destroy(o);
      throw;
    }
    o->finalized = true;
    *coarse = o;
  });
}

int pa_op_apply_add(pa_op *op, const double *x, double *y, void *stream) {
  return guarded([&] { apply(op, x, y, false, (hipStream_t)stream); });
}

int pa_op_mult(pa_op *op, const double *x, double *y, void *stream) {
  return guarded([&] {
    apply(op, x, y, true, (hipStream_t)stream);
  });
}

/* A^T: same trial and test evaluation, so A^T = E^T B^T D(C^T) B E -- the forward kernels with every coefficient matrix
 * transposed (the reference builds a second libCEED operator with trial and test swapped, fem/libceed/operator.cpp:
 * 199-240; for symmetric coefficients this is the forward apply, like its SymmetricOperator wrapper). */
int pa_op_mult_transpose(pa_op *op, const double *x, double *y, void *stream) {
  return guarded([&] {
    // (two-space operators: x has the size of the test space, y of the trial space)
    PA_REQUIRE(op && (op->height == op->width || (op->subs.empty() && op->dsubs.empty())), "transpose apply needs a square operator");
    TransposeScope t(!op->symmetric());
    apply(op, x, y, true, (hipStream_t)stream);
  });
}

int pa_op_apply_add_transpose(pa_op *op, const double *x, double *y, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && (op->height == op->width || (op->subs.empty() && op->dsubs.empty())), "transpose apply needs a square operator");
    TransposeScope t(!op->symmetric());
    apply(op, x, y, false, (hipStream_t)stream);
  });
}

/* 0: no fused form; 1: hexahedral streaming form; 2: dense-table form; 3: hexahedral form followed by further (surface)
 * sub-operators.  In forms 2 and 3 the FIRST sub-operators of the two parts pair up in one pass over the element data; every further
 * dense sub-operator of either part (the surface terms of a driven problem -- absorbing boundary, lumped ports, the A2(omega)
 * terms of spaceoperator.cpp:786-804: a handful of boundary faces) is applied after it on both parts of x. */
int pa_op_complex_fused(const pa_op *op_r, const pa_op *op_i) {
  if (!op_r || !op_i || !op_r->msubs.empty() || !op_i->msubs.empty() || op_r->height != op_i->height || op_r->width != op_i->width)
    return 0;
  const bool hex = op_r->subs.size() == 1 && op_i->subs.size() == 1;
  const bool dense = op_r->dsubs.size() >= 1 && op_i->dsubs.size() >= 1 && op_r->subs.empty() && op_i->subs.empty();
  if (!hex && !dense) return 0;
  // (the check compares the two restrictions on the host: once per pair)
  if (op_r->cplx_partner != op_i->id) {
    op_r->cplx_ok = hex ? (nd_hex_stream_complex_ok(*op_r->subs[0], *op_i->subs[0]) ? ((op_r->dsubs.empty() && op_i->dsubs.empty()) ? 1 : 3) : 0)
                        : (dense_complex_ok(*op_r->dsubs[0], *op_i->dsubs[0]) ? 2 : 0);
    op_r->cplx_partner = op_i->id;
  }
  return op_r->cplx_ok;
}

int pa_op_mult_complex(pa_op *op_r, pa_op *op_i, const double *xr, const double *xi, double *yr, double *yi, int ess_policy,
                       void *stream) {
  return guarded([&] {
    PA_REQUIRE(op_r && op_i && xr && xi && yr && yi, "null argument");
    const int kind = pa_op_complex_fused(op_r, op_i);
    PA_REQUIRE(kind, "the two operators have no fused complex form (pa_op_complex_fused)");
    PA_REQUIRE(xr != yr && xr != yi && xi != yr && xi != yi, "in-place apply is not supported");
    const bool masked = ess_policy >= 0;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 2) {  // dense tables (tetrahedra, ...)
      DenseSub *dr = op_r->dsubs[0];
      // ParOperator's essential dofs (round 5): entries read as zero through the flagged index copy of the REAL operator, rows
      // fixed by the two gathers -- yr[ess] = xr[ess] | 0, yi[ess] = xi[ess] | 0 (rap.cpp:450-457 on one rank, as the hex form)
      PA_REQUIRE(!masked || (op_r->has_essential && dr->d_idx_bc && dr->d_ess_flag),
                 "pa_op_set_essential has not been called on the real operator");
      if (!dr->d_ye2) dr->d_ye2 = dev_alloc<double>((size_t)dr->nb * dr->KP * 64);
      launch_dense_complex(*dr, *op_i->dsubs[0], xr, xi, dr->d_ye2, s, masked);
      launch_dense_gather(*dr, yr, false, s, nullptr, nullptr, masked ? xr : nullptr, ess_policy);
      launch_dense_gather(*dr, yi, false, s, dr->d_ye2, nullptr, masked ? xi : nullptr, ess_policy);
    } else {
      SubOp *sr = op_r->subs[0];
      PA_REQUIRE(!masked || (op_r->has_essential && sr->d_perm_s_bc), "pa_op_set_essential has not been called on the real operator");
      if (!sr->d_ye2) sr->d_ye2 = dev_alloc<double>((size_t)((sr->ne + 3) & ~3) * sr->P);
      if (sr->qd->metric) stream_element_coefficients(*op_i->subs[0]);
      launch_nd_hex_stream_complex(*sr, *op_i->subs[0], xr, xi, yr, yi, sr->d_ye2, masked, s);
      // (q1d = 5: the wide kernel's gather keeps its two launches)
      static const bool one_gather = !(getenv("PALACE_AMD_CPLX_GATHER2") && atoi(getenv("PALACE_AMD_CPLX_GATHER2")) == 0);
      if (one_gather) {  // both parts in one launch: headers and copy positions read once (round 6)
        launch_et_run_gather2(*sr, yr, yi, s, xr, xi, masked, ess_policy, sr->d_ye2);
      } else {
        launch_et_run_gather(*sr, yr, false, s, xr, masked, ess_policy);
        launch_et_run_gather(*sr, yi, false, s, xi, masked, ess_policy, sr->d_ye2);
      }
    }
    // the remaining sub-operators (operator.cpp:98-134 term by term): B of the real part, yr += B xr, yi += B xi; B of the
    // imaginary part, yi += B xr, yr -= B xi; essential entries of x read as zero through B's own flagged index copy, essential
    // rows of y left as the gathers above fixed them
    for (int part = 0; part < 2; part++) {
      const pa_op *op = part ? op_i : op_r;
      for (size_t k = (kind == 2 ? 1 : 0); k < op->dsubs.size(); k++) {
        const DenseSub &b = *op->dsubs[k];
        PA_REQUIRE(!masked || (op->has_essential && b.d_ess_flag),
                   part ? "pa_op_set_essential has not been called on the imaginary operator"
                        : "pa_op_set_essential has not been called on the real operator");
        launch_dense_apply(b, xr, masked, s);
        launch_dense_gather_signed(b, part ? yi : yr, +1.0, masked, s);
        launch_dense_apply(b, xi, masked, s);
        launch_dense_gather_signed(b, part ? yr : yi, part ? -1.0 : +1.0, masked, s);
      }
    }
  });
}

int pa_op_is_symmetric(const pa_op *op) { return (op && op->symmetric()) ? 1 : 0; }
int pa_op_dense_affine(const pa_op *op) {
  int n = 0;
  if (op)
    for (const DenseSub *ds : op->dsubs) n += ds->d_affine ? 1 : 0;
  return n;
}
int pa_op_streams(const pa_op *op) {
  return (op && op->subs.size() == 1 && op->dsubs.empty() && op->subs[0]->d_idxc && op->subs[0]->stream_default) ? 1 : 0;
}

/* 0: no essential list set on this operator, 1: this list is set, -1: a different one is */
int pa_op_essential_state(const pa_op *op, const int32_t *ess, int32_t n) {
  if (!op || !op->has_essential) return 0;
  std::vector<int32_t> v(ess, ess + (ess ? n : 0));
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
  return v == op->ess_sorted ? 1 : -1;
}

int pa_op_set_essential(pa_op *op, const int32_t *ess, int32_t n) {
  return guarded([&] {
    PA_REQUIRE(op && (ess || n == 0), "null argument");
    // the flagged index tables live in the operator: a second wrapper with another list must not overwrite them under
    // the first one's feet (it takes the unfused path instead, linalg.hip ParOperator)
    PA_REQUIRE(pa_op_essential_state(op, ess, n) >= 0,
               "a different essential dof list is already fused into this operator (pa_op_essential_state)");
    op->ess_sorted.assign(ess, ess + n);
    std::sort(op->ess_sorted.begin(), op->ess_sorted.end());
    op->ess_sorted.erase(std::unique(op->ess_sorted.begin(), op->ess_sorted.end()), op->ess_sorted.end());
    std::vector<char> flag((size_t)op->width, 0);
    for (int i = 0; i < n; i++) {
      PA_REQUIRE(ess[i] >= 0 && ess[i] < op->width, "essential dof out of range");
      flag[ess[i]] = 1;
    }
    for (SubOp *so : op->subs) {
      std::vector<int32_t> bc(so->h_sidx);
      for (auto &s : bc) {
        const int d = s >= 0 ? s : -1 - s;
        if (flag[d]) s = s >= 0 ? (d | kEssBit) : -1 - (d | kEssBit);
      }
      hipFree(so->d_sidx_bc);
      so->d_sidx_bc = dev_upload(bc.data(), bc.size());
    }
    for (DenseSub *ds : op->dsubs) dense_set_essential(*ds, flag);
    for (SubOp *so : op->subs) {
      if (so->d_shared) {  // flagged copy of the gather list: essential rows are fixed up inside the gather kernel
        std::vector<int32_t> lb(so->h_shared);
        for (auto &d : lb)
          if (flag[d]) d |= kEssBit;
        hipFree(so->d_shared_bc);
        so->d_shared_bc = dev_upload(lb.data(), lb.size());
      }
      if (so->d_idxc) stream_set_essential(*so, flag);
    }
    op->has_essential = true;
  });
}

int pa_op_set_interface_dofs(pa_op *op, const int32_t *ldofs, int32_t n) {
  return guarded([&] {
    PA_REQUIRE(op && (ldofs || n == 0), "null argument");
    std::vector<char> flag((size_t)op->width, 0);
    for (int i = 0; i < n; i++) {
      PA_REQUIRE(ldofs[i] >= 0 && ldofs[i] < op->width, "interface dof out of range");
      flag[ldofs[i]] = 1;
    }
    for (SubOp *so : op->subs) stream_set_interface(*so, flag);
  });
}

int pa_op_mult_after(pa_op *op, const double *x, double *y, void *stream, void *event) {
  return guarded([&] { apply(op, x, y, true, (hipStream_t)stream, false, -1, (hipEvent_t)event); });
}

int pa_op_mult_essential(pa_op *op, const double *x, double *y, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && op->has_essential, "pa_op_set_essential has not been called");
    apply(op, x, y, true, (hipStream_t)stream, true);
  });
}

int pa_op_mult_essential_diag(pa_op *op, const double *x, double *y, int diag_policy, void *stream, int *handled) {
  return guarded([&] {
    PA_REQUIRE(op && op->has_essential && handled, "pa_op_set_essential has not been called");
    if (op->subs.empty() && op->msubs.empty() && op->dsubs.size() == 1 && op->dsubs[0]->d_ess_flag && x != y &&
        !TransposeScope::active()) {
      // one dense-table block: essential entries read as zero through the flagged index copy, rows fixed by the gather
      const DenseSub &ds = *op->dsubs[0];
      launch_dense_apply(ds, x, true, (hipStream_t)stream);
      launch_dense_gather(ds, y, false, (hipStream_t)stream, nullptr, nullptr, x, diag_policy ? 1 : 0);
      *handled = 1;
      return;
    }
    const bool fuse = op->subs.size() == 1 && op->dsubs.empty() &&
                      (op->subs[0]->fe_type == PA_FE_HCURL ? nd_hex_fuses_essential(*op->subs[0])
                                                           : (op->subs[0]->d_idxc && op->subs[0]->stream_default && op->subs[0]->d_perm_s_bc));
    apply(op, x, y, true, (hipStream_t)stream, true, fuse ? (diag_policy ? 1 : 0) : -1);
    *handled = fuse ? 1 : 0;
  });
}

int pa_op_prepare_fused_step(pa_op *op, int *available) {
  return guarded([&] {
    PA_REQUIRE(op && available, "null argument");
    *available = 0;
    const bool enabled = !(getenv("PALACE_AMD_FUSED_STEP") && atoi(getenv("PALACE_AMD_FUSED_STEP")) == 0);
    if (!enabled || !op->has_essential || !op->msubs.empty()) return;
    if (op->subs.empty() && op->dsubs.size() == 1) {  // one dense-table block: its CSR-form gather owns every row already
      *available = dense_fused_step_ok(*op->dsubs[0]) ? 1 : 0;
      return;
    }
    if (op->subs.empty() && op->dsubs.size() > 1) {
      // a volume block and row-limited surface blocks (the absorbing boundary and the lumped ports of a driven problem: a few
      // thousand faces in a space of millions): the surface blocks accumulate into a side vector first, the volume block's gather
      // adds it and runs the step.  PALACE_AMD_FUSED_STEP_SURFACE=0: off.
      const char *e = getenv("PALACE_AMD_FUSED_STEP_SURFACE");
      if ((e && e[0] == '0') || !dense_fused_step_ok(*op->dsubs[0]) || op->dsubs[0]->d_rows) return;
      for (size_t k = 1; k < op->dsubs.size(); k++)
        if (!op->dsubs[k]->d_rows || !op->dsubs[k]->d_idx_bc) return;
      if (!op->d_t_extra) {
        std::vector<int32_t> rows;
        for (size_t k = 1; k < op->dsubs.size(); k++) {
          std::vector<int32_t> r((size_t)op->dsubs[k]->n_rows);
          if (!r.empty()) PA_HIP(hipMemcpy(r.data(), op->dsubs[k]->d_rows, r.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
          rows.insert(rows.end(), r.begin(), r.end());
        }
        std::sort(rows.begin(), rows.end());
        rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
        op->n_extra_rows = (int)rows.size();
        op->d_extra_rows = dev_upload(rows.data(), std::max<size_t>(rows.size(), 1));
        if (op->dsubs.size() - 1 <= (size_t)kMaxSurfaceBlocks) {  // one gather for all the surface blocks (else: one each, accumulating)
          std::vector<int32_t> uptr(rows.size() + 1, 0), uent;
          std::vector<uint8_t> ublk;
          std::vector<std::vector<int32_t>> tptr(op->dsubs.size()), tent(op->dsubs.size());
          for (size_t k = 1; k < op->dsubs.size(); k++) {
            const DenseSub &d = *op->dsubs[k];
            tptr[k].resize((size_t)d.lsize + 1), tent[k].resize((size_t)d.ne * d.P);
            PA_HIP(hipMemcpy(tptr[k].data(), d.d_tptr, tptr[k].size() * sizeof(int32_t), hipMemcpyDeviceToHost));
            PA_HIP(hipMemcpy(tent[k].data(), d.d_tent, tent[k].size() * sizeof(int32_t), hipMemcpyDeviceToHost));
          }
          for (size_t i = 0; i < rows.size(); i++) {
            for (size_t k = 1; k < op->dsubs.size(); k++)
              for (int32_t a = tptr[k][(size_t)rows[i]]; a < tptr[k][(size_t)rows[i] + 1]; a++) uent.push_back(tent[k][(size_t)a]), ublk.push_back((uint8_t)(k - 1));
            uptr[i + 1] = (int32_t)uent.size();
          }
          op->d_urow_ptr = dev_upload(uptr.data(), uptr.size());
          op->d_uent = dev_upload(uent.data(), std::max<size_t>(uent.size(), 1));
          op->d_ublk = dev_upload(ublk.data(), std::max<size_t>(ublk.size(), 1));
        }
        op->d_t_extra = dev_alloc<double>((size_t)op->height);
        PA_HIP(hipMemset(op->d_t_extra, 0, sizeof(double) * (size_t)op->height));
      }
      *available = 1;
      return;
    }
    if (op->subs.size() != 1 || !op->dsubs.empty()) return;
    SubOp *so = op->subs[0];
    // (four points per direction: pa_nd_hex_stream.hip; five: pa_nd_hex_stream5.hip -- nd_hex_stream_ok covers both; H1 blocks
    // whose y = A x runs on the streaming kernel: pa_h1_hex_stream.hip)
    if (so->fe_type == PA_FE_H1) {
      if (!so->d_idxc || !so->stream_default || !so->d_perm_s_bc || !h1_hex_stream_ok(*so)) return;
    } else if (so->fe_type != PA_FE_HCURL || (so->q1d != 4 && so->q1d != 5) || !so->d_idxc || !so->d_perm_s_bc || !nd_hex_stream_ok(*so)) {
      return;
    }
    *available = stream_build_all(*so) ? 1 : 0;
  });
}

// the surface blocks of a dense operator prepared for the fused step: their contributions to t_extra (pa_op_prepare_fused_step)
static const double *dense_surface_terms(pa_op *op, const double *x, hipStream_t s) {
  if (op->dsubs.size() < 2) return nullptr;
  PA_REQUIRE(op->d_t_extra, "pa_op_prepare_fused_step has not been called (or found no fused form)");
  if (op->d_urow_ptr) {  // every surface block's element kernel, then ONE gather over the union of their rows (it overwrites them)
    SurfaceYe ye{};
    for (size_t k = 1; k < op->dsubs.size(); k++) {
      launch_dense_apply(*op->dsubs[k], x, true, s);
      ye.ye[k - 1] = op->dsubs[k]->d_ye;
    }
    launch_surface_rows(op->d_extra_rows, op->n_extra_rows, op->d_urow_ptr, op->d_uent, op->d_ublk, ye, op->d_t_extra, s);
    return op->d_t_extra;
  }
  launch_zero_rows(op->d_t_extra, op->d_extra_rows, op->n_extra_rows, s);
  for (size_t k = 1; k < op->dsubs.size(); k++) {
    launch_dense_apply(*op->dsubs[k], x, true, s);
    launch_dense_gather(*op->dsubs[k], op->d_t_extra, true, s);
  }
  return op->d_t_extra;
}

int pa_op_mult_cheb_step(pa_op *op, const double *x, const pa_cheb_step *step, int diag_policy, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && x && step && step->dinv && step->r0 && step->out, "null argument");
    PA_REQUIRE(x != step->out, "the step cannot overwrite its own input");
    PA_REQUIRE(!TransposeScope::active() || op->symmetric(), "transposed step of a non-symmetric operator");
    if (op->subs.empty() && !op->dsubs.empty() && dense_fused_step_ok(*op->dsubs[0])) {
      const DenseSub &ds = *op->dsubs[0];
      const double *t_add = dense_surface_terms(op, x, (hipStream_t)stream);
      launch_dense_apply(ds, x, true, (hipStream_t)stream);
      launch_dense_gather_step(ds, x, GatherStep{step->sd, step->sr, step->dinv, step->r0, step->e_prev, step->out, step->add, nullptr, 1,
                                                nullptr, nullptr, t_add},
                               diag_policy ? 1 : 0, (hipStream_t)stream);
      return;
    }
    PA_REQUIRE(op->subs.size() == 1 && op->subs[0]->n_all > 0, "pa_op_prepare_fused_step has not been called (or found no fused form)");
    const SubOp &so = *op->subs[0];
    launch_nd_hex_stream_all(so, x, (hipStream_t)stream);
    launch_et_run_gather_step(so, x, GatherStep{step->sd, step->sr, step->dinv, step->r0, step->e_prev, step->out, step->add, nullptr, 1},
                              diag_policy ? 1 : 0, (hipStream_t)stream);
  });
}

int pa_op_mult_residual(pa_op *op, const double *y, const double *b, double *res, const double *dinv, double c0, double *d0,
                        int diag_policy, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && y && b && (res || d0) && (!d0 || dinv), "null argument");
    PA_REQUIRE(y != res && y != d0, "the residual cannot overwrite the operator's input");
    PA_REQUIRE(!TransposeScope::active() || op->symmetric(), "transposed step of a non-symmetric operator");
    if (op->subs.empty() && !op->dsubs.empty() && dense_fused_step_ok(*op->dsubs[0])) {
      const DenseSub &ds = *op->dsubs[0];
      const double *t_add = dense_surface_terms(op, y, (hipStream_t)stream);
      launch_dense_apply(ds, y, true, (hipStream_t)stream);
      launch_dense_gather_step(ds, y, GatherStep{0.0, c0, dinv, b, nullptr, d0, 0, res, 2, nullptr, nullptr, t_add}, diag_policy ? 1 : 0,
                               (hipStream_t)stream);
      return;
    }
    PA_REQUIRE(op->subs.size() == 1 && op->subs[0]->n_all > 0, "pa_op_prepare_fused_step has not been called (or found no fused form)");
    const SubOp &so = *op->subs[0];
    launch_nd_hex_stream_all(so, y, (hipStream_t)stream);
    launch_et_run_gather_step(so, y, GatherStep{0.0, c0, dinv, b, nullptr, d0, 0, res, 2}, diag_policy ? 1 : 0, (hipStream_t)stream);
  });
}

int pa_op_supports_split(const pa_op *op) {
  if (!op || !op->msubs.empty()) return 0;
  if (op->subs.size() == 1 && op->dsubs.empty()) {
    const SubOp &so = *op->subs[0];
    return (nd_hex_stream_split_ok(so) || (so.fe_type == PA_FE_H1 && so.d_idxc)) ? 1 : 0;
  }
  // dense-table blocks (tetrahedra, ...): one block on the resident kernel
  if (op->subs.empty() && op->dsubs.size() == 1) return dense_split_ok(*op->dsubs[0]) ? 1 : 0;
  return 0;
}

int pa_op_mult_split(pa_op *op, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *y,
                     double *yg, int n_true, int ess_policy, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && x && y && xg0 && yg, "null argument");
    PA_REQUIRE(pa_op_supports_split(op), "the operator has no split-vector form (pa_op_supports_split)");
    PA_REQUIRE(x != y, "in-place apply is not supported");
    PA_REQUIRE(!TransposeScope::active() || op->symmetric(), "transposed split apply of a non-symmetric operator");
    const bool masked = ess_policy >= 0;
    const SplitIO io{n_true, xg0, xg1, sel, yg};
    hipStream_t s = (hipStream_t)stream;
    if (!op->dsubs.empty()) {  // dense-table block: the essential entries are flagged in the masked index copy, the rows are
                               // fixed by its gather
      DenseSub *ds = op->dsubs[0];
      PA_REQUIRE(!masked || op->has_essential, "pa_op_set_essential has not been called");
      launch_dense_apply(*ds, x, masked, s, &io);
      launch_dense_gather(*ds, y, false, s, nullptr, &io, x, ess_policy);
      return;
    }
    SubOp *so = op->subs[0];
    PA_REQUIRE(!masked || (op->has_essential && so->d_perm_s_bc), "pa_op_set_essential has not been called");
    if (so->fe_type == PA_FE_H1)
      launch_h1_hex_stream(*so, x, y, masked, s, &io);
    else
      launch_nd_hex_stream(*so, x, y, masked, s, -1, &io);
    launch_et_run_gather(*so, y, false, s, x, masked, ess_policy, nullptr, &io);
  });
}

int pa_op_mult_split_step(pa_op *op, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *yg,
                          int n_true, int ess_policy, const pa_split_step *st, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && x && xg0 && yg && st && st->r0 && st->iface_mask && st->t_iface, "null argument");
    PA_REQUIRE(st->mode == 1 ? (st->dinv && st->out) : (st->mode == 2 && (st->res || st->out) && (!st->out || st->dinv)), "invalid step");
    PA_REQUIRE(x != st->out && x != st->res, "the step cannot overwrite the operator's input");
    PA_REQUIRE(!TransposeScope::active() || op->symmetric(), "transposed split apply of a non-symmetric operator");
    PA_REQUIRE(ess_policy < 0 || op->has_essential, "pa_op_set_essential has not been called");
    const SplitIO io{n_true, xg0, xg1, sel, yg};
    hipStream_t s = (hipStream_t)stream;
    if (op->subs.empty() && op->dsubs.size() == 1 && dense_fused_step_ok(*op->dsubs[0]) && dense_split_ok(*op->dsubs[0])) {
      const DenseSub &ds = *op->dsubs[0];
      launch_dense_apply(ds, x, ess_policy >= 0, s, &io);
      launch_dense_gather_step(ds, x, GatherStep{st->sd, st->sr, st->dinv, st->r0, st->e_prev, st->out, st->add, st->res, st->mode,
                                                 st->iface_mask, st->t_iface}, ess_policy, s, &io);
      return;
    }
    PA_REQUIRE(op->subs.size() == 1 && op->dsubs.empty() && op->subs[0]->n_all > 0,
               "pa_op_prepare_fused_step has not been called (or found no fused form)");
    const SubOp &so = *op->subs[0];
    launch_nd_hex_stream_all(so, x, s, &io);
    launch_et_run_gather_step(so, x, GatherStep{st->sd, st->sr, st->dinv, st->r0, st->e_prev, st->out, st->add, st->res, st->mode,
                                                st->iface_mask, st->t_iface}, ess_policy, s, &io);
  });
}

int pa_op_mult2(pa_op *op, const double *x0, const double *x1, double *y0, double *y1, void *stream) {
  return guarded([&] { apply2(op, x0, x1, y0, y1, (hipStream_t)stream, false, -1); });
}

int pa_op_mult2_essential_diag(pa_op *op, const double *x0, const double *x1, double *y0, double *y1, int diag_policy,
                               void *stream, int *handled) {
  return guarded([&] {
    PA_REQUIRE(op && op->has_essential && handled, "pa_op_set_essential has not been called");
    *handled = apply2(op, x0, x1, y0, y1, (hipStream_t)stream, true, diag_policy ? 1 : 0) ? 1 : 0;
  });
}

int pa_op_assemble_diagonal(pa_op *op, double *diag, void *stream) {
  return guarded([&] {
    PA_REQUIRE(op && diag, "null argument");
    PA_REQUIRE(op->msubs.empty(), "mixed-space operators have no diagonal");
    PA_HIP(hipMemsetAsync(diag, 0, sizeof(double) * (size_t)op->height, (hipStream_t)stream));
    for (const SubOp *so : op->subs) {
      if (so->fe_type == PA_FE_HCURL)
        launch_nd_hex_diag(*so, diag, (hipStream_t)stream);
      else
        launch_h1_hex_diag(*so, diag, (hipStream_t)stream);
    }
    for (const DenseSub *ds : op->dsubs) launch_dense_diag(*ds, diag, (hipStream_t)stream);
  });
}

int pa_op_stream_affine(const pa_op *op, int32_t out[3]) {
  return guarded([&] {
    PA_REQUIRE(op && out, "null argument");
    out[0] = out[1] = out[2] = 0;
    for (const SubOp *so : op->subs)
      if (so->fe_type == PA_FE_HCURL && so->qd) {
        out[0] = so->ne, out[1] = so->qd->n_aff_elems, out[2] = so->qd->d_aff ? so->qd->n_aff_batch_elems : 0;
        return;
      }
  });
}
int pa_op_dense_gather_form(const pa_op *op, int32_t out[2]) {
  return guarded([&] {
    PA_REQUIRE(op && out, "null argument");
    out[0] = out[1] = 0;
    if (!op->dsubs.empty()) out[0] = op->dsubs[0]->ye_rows ? 1 : 0, out[1] = dense_gather_group(*op->dsubs[0]);
  });
}
int pa_op_num_sub(const pa_op *op) { return op ? (int)(op->subs.size() + op->dsubs.size() + op->msubs.size()) : -1; }
int pa_op_destroy_assembly_data(const pa_op *op) {
  return guarded([&] { PA_REQUIRE(op, "null operator"); });
}
int pa_op_height(const pa_op *op) { return op ? op->height : -1; }
int pa_op_width(const pa_op *op) { return op ? op->width : -1; }

double pa_op_algorithmic_bytes(const pa_op *op) {
  if (!op) return 0.0;
  double bytes = 0.0;
  // (o = 1 orientation byte per entry for the oriented restriction of H(curl) blocks, none for H1: SURVEY.md 8d)
  for (const SubOp *so : op->subs) bytes += (double)so->ne * ((double)so->Q * 11 * 8 + (double)so->P * (so->fe_type == PA_FE_H1 ? 4 : 5));
  for (const DenseSub *ds : op->dsubs)  // o = 3 for the curl-oriented restriction; G = 11 (3-D) or 6 (2-D)
    bytes += (double)ds->ne * ((double)ds->Q * ds->geom->nrows * 8 + (double)ds->P * (ds->d_co ? 7 : 5));
  return bytes + 16.0 * op->height;
}

void pa_op_destroy(pa_op *op) {
  if (!op) return;
  for (SubOp *so : op->subs) free_sub(so);
  for (DenseSub *ds : op->dsubs) free_dense_sub(ds);
  for (MixedSub *ms : op->msubs) free_mixed_sub(ms);
  (void)hipFree(op->d_t_extra), (void)hipFree(op->d_extra_rows);
  (void)hipFree(op->d_urow_ptr), (void)hipFree(op->d_uent), (void)hipFree(op->d_ublk);
  delete op;
}

}  // extern "C"
