// palace::fem front end over the C ABI (see fem.hpp for the reference symbols each class follows).  Host code only.
#include "fem.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace palace {

Operator *make_interp_operator(const Context &ctx, const pa_restriction_desc &rc, const pa_basis_desc &bc,
                               const pa_restriction_desc &rf, const pa_basis_desc &bf, const double *Ic,
                               const double *Io, const Halo *halo_c, int nt_c, int nt_f, int kind);

Operator *make_dense_interp_operator(const Context &ctx, const pa_restriction_desc &rd, const pa_restriction_desc &rr,
                                     const double *M, const Halo *halo_d, int nt_d, int nt_r, int nmat, const uint8_t *mat_id);

namespace {
void check(int rc) {
  if (rc) throw pa::Error(pa_last_error());
}
}  // namespace

// ---- MaterialPropertyCoefficient (materialoperator.cpp:586-868) -------------------------------------------------------
MaterialPropertyCoefficient::MaterialPropertyCoefficient(int attr_max) : attr_mat_((size_t)attr_max, -1) {}

MaterialPropertyCoefficient::MaterialPropertyCoefficient(const std::vector<int> &attr_mat, int dim,
                                                         const std::vector<double> &mat_coeff, double a)
    : attr_mat_(attr_mat), dim_(dim), mat_coeff_(mat_coeff) {
  PA_REQUIRE(dim > 0 && mat_coeff.size() % ((size_t)dim * dim) == 0, "Invalid dimensions for MaterialPropertyCoefficient!");
  num_mat_ = (int)(mat_coeff.size() / ((size_t)dim * dim));
  *this *= a;
}

void MaterialPropertyCoefficient::Resize(int dim, int num_mat) {
  // keeps the existing matrices; a change 1 -> dim spreads the scalars over the diagonal
  std::vector<double> next((size_t)num_mat * dim * dim, 0.0);
  for (int k = 0; k < std::min(num_mat, num_mat_); k++) {
    if (dim == dim_) {
      std::copy(Mat(k), Mat(k) + dim * dim, next.begin() + (size_t)k * dim * dim);
    } else if (dim_ == 1) {
      for (int i = 0; i < dim; i++) next[(size_t)k * dim * dim + i * (dim + 1)] = Mat(k)[0];
    }
  }
  mat_coeff_.swap(next);
  dim_ = dim, num_mat_ = num_mat;
}

void MaterialPropertyCoefficient::UpdateProperty(int k, const double *coeff, int cdim, double a) {
  if (dim_ == 0) {  // first material: takes the shape of the coefficient
    PA_REQUIRE(k == 0 && num_mat_ == 1, "Unexpected initial size for MaterialPropertyCoefficient!");
    dim_ = cdim;
    mat_coeff_.assign((size_t)cdim * cdim, 0.0);
    for (int i = 0; i < cdim * cdim; i++) Mat(0)[i] = a * coeff[i];
  } else if (cdim == dim_) {
    for (int i = 0; i < cdim * cdim; i++) Mat(k)[i] += a * coeff[i];
  } else if (cdim == 1) {  // add as diagonal
    for (int i = 0; i < dim_; i++) Mat(k)[i * (dim_ + 1)] += a * coeff[0];
  } else if (dim_ == 1) {  // convert to matrix coefficients, the previous scalars become diagonals
    Resize(cdim, num_mat_);
    for (int i = 0; i < cdim * cdim; i++) Mat(k)[i] += a * coeff[i];
  } else {
    throw pa::Error("Invalid dimensions when updating material property!");
  }
}

bool MaterialPropertyCoefficient::Equals(int k, const double *coeff, int cdim, double a) const {
  constexpr double tol = 1.0e-9;
  const double *m = Mat(k);
  if (cdim == 1) {
    for (int i = 0; i < dim_; i++) {
      if (std::abs(m[i * (dim_ + 1)] - a * coeff[0]) >= tol * std::abs(m[i * (dim_ + 1)])) return false;
      for (int j = 0; j < dim_; j++)
        if (j != i && std::abs(m[i + dim_ * j]) > 0.0) return false;
    }
    return true;
  }
  if (cdim != dim_) return false;
  double diff = 0.0, ref = 0.0;
  for (int i = 0; i < dim_ * dim_; i++) diff = std::max(diff, std::abs(m[i] - a * coeff[i])), ref = std::max(ref, std::abs(m[i]));
  return diff < tol * ref;
}

void MaterialPropertyCoefficient::AddCoefficient(const std::vector<int> &attr_mat, int dim, const std::vector<double> &mat_coeff,
                                                 double a) {
  const int nk = dim > 0 ? (int)(mat_coeff.size() / ((size_t)dim * dim)) : 0;
  if (empty()) {
    PA_REQUIRE(attr_mat.size() == attr_mat_.size(),
               "Invalid resize of attribute to material property map in MaterialPropertyCoefficient::AddCoefficient!");
    attr_mat_ = attr_mat, dim_ = dim, mat_coeff_ = mat_coeff, num_mat_ = nk;
    *this *= a;
  } else if (attr_mat == attr_mat_) {
    PA_REQUIRE(nk == num_mat_, "Invalid dimensions for MaterialPropertyCoefficient::AddCoefficient!");
    for (int k = 0; k < nk; k++) UpdateProperty(k, mat_coeff.data() + (size_t)k * dim * dim, dim, a);
  } else {
    for (int k = 0; k < nk; k++) {
      std::vector<int> attr_list;
      for (size_t i = 0; i < attr_mat.size(); i++)
        if (attr_mat[i] == k) attr_list.push_back((int)i + 1);
      AddMaterialProperty(attr_list, mat_coeff.data() + (size_t)k * dim * dim, dim, a);
    }
  }
}

void MaterialPropertyCoefficient::AddMaterialProperty(const std::vector<int> &attr_list, const double *coeff, int dim, double a) {
  // attributes that already have a material must all point to the same one, which is then updated in place;
  // otherwise an equal existing material is reused or a new one appended
  if (attr_list.empty()) return;
  int mat_idx = -1;
  bool first = true;
  for (int attr : attr_list) {
    PA_REQUIRE(attr >= 1 && attr <= (int)attr_mat_.size(),
               "Out of bounds access for attribute in MaterialPropertyCoefficient::AddMaterialProperty!");
    if (first)
      mat_idx = attr_mat_[attr - 1], first = false;
    else
      PA_REQUIRE(mat_idx == attr_mat_[attr - 1],
                 "All attributes for MaterialPropertyCoefficient::AddMaterialProperty must correspond to the same existing "
                 "material if it exists!");
  }
  if (mat_idx < 0) {
    for (int k = 0; k < num_mat_ && dim_ > 0; k++)
      if (Equals(k, coeff, dim, a)) {
        mat_idx = k;
        break;
      }
    if (mat_idx < 0) {
      if (dim_ == 0)
        num_mat_ += 1;  // shape fixed by UpdateProperty below
      else
        Resize(dim_, num_mat_ + 1);
      mat_idx = num_mat_ - 1;
    }
    if (dim_ > 0) std::fill(Mat(mat_idx), Mat(mat_idx) + dim_ * dim_, 0.0);  // zero out so we can add
    for (int attr : attr_list) attr_mat_[attr - 1] = mat_idx;
  }
  UpdateProperty(mat_idx, coeff, dim, a);
}

MaterialPropertyCoefficient &MaterialPropertyCoefficient::operator*=(double a) {
  for (double &v : mat_coeff_) v *= a;
  return *this;
}

void MaterialPropertyCoefficient::RestrictCoefficient(const std::vector<int> &attr_list) {
  const std::vector<int> attr_orig(attr_mat_);
  const std::vector<double> coeff_orig(mat_coeff_);
  const int d2 = dim_ * dim_;
  std::fill(attr_mat_.begin(), attr_mat_.end(), -1);
  mat_coeff_.clear(), num_mat_ = 0;
  for (int attr : attr_list) {
    if (attr_mat_[attr - 1] >= 0) continue;  // already processed
    const int orig = attr_orig[attr - 1], idx = num_mat_;
    for (int attr2 : attr_list)
      if (attr_orig[attr2 - 1] == orig) attr_mat_[attr2 - 1] = idx;
    mat_coeff_.insert(mat_coeff_.end(), coeff_orig.begin() + (size_t)orig * d2, coeff_orig.begin() + (size_t)(orig + 1) * d2);
    num_mat_++;
  }
}

void MaterialPropertyCoefficient::NormalProjectedCoefficient(const std::array<double, 3> &normal) {
  std::vector<double> next((size_t)num_mat_);
  for (int k = 0; k < num_mat_; k++) {
    double s = 0.0;
    for (int i = 0; i < dim_; i++)
      for (int j = 0; j < dim_; j++) s += normal[i] * Mat(k)[i + dim_ * j] * normal[j];
    next[k] = s;
  }
  mat_coeff_.swap(next), dim_ = 1;
}

// ---- coefficient context (coefficient.cpp:51-131) -------------------------------------------------------------------
namespace ceed {

namespace {
void set_int(double &slot, int v) {
  int64_t w = 0;
  std::memcpy(&w, &slot, 8);
  w = (w & ~0xffffffffll) | (uint32_t)v;
  std::memcpy(&slot, &w, 8);
}
}  // namespace

std::vector<double> PopulateCoefficientContext(int dim, const MaterialPropertyCoefficient *Q, bool transpose, double a) {
  const int d2 = dim * dim;
  if (!Q) {  // no attribute map: every attribute uses the identity scaled by a
    std::vector<double> ctx(2 + (size_t)d2, 0.0);
    set_int(ctx[0], 0), set_int(ctx[1], 1);
    for (int i = 0; i < dim; i++) ctx[2 + (size_t)i * (dim + 1)] = a;
    return ctx;
  }
  const auto &attr_mat = Q->GetAttributeToMaterial();
  const int nattr = (int)attr_mat.size(), nmat = Q->NumMaterials(), qd = Q->Dimension();
  PA_REQUIRE(nattr > 0, "Empty attributes for MaterialPropertyCoefficient!");
  PA_REQUIRE(*std::max_element(attr_mat.begin(), attr_mat.end()) < nmat, "Invalid attribute material property for MaterialPropertyCoefficient!");
  PA_REQUIRE(nmat == 0 || qd == 1 || qd == dim, "Dimension mismatch for MaterialPropertyCoefficient and libCEED integrator!");
  // unassigned attributes map to the zero coefficient stored after the last material
  std::vector<double> ctx(2 + (size_t)nattr + (size_t)d2 * (nmat + 1), 0.0);
  set_int(ctx[0], nattr);
  for (int i = 0; i < nattr; i++) set_int(ctx[1 + (size_t)i], attr_mat[i] < 0 ? nmat : attr_mat[i]);
  set_int(ctx[1 + (size_t)nattr], nmat + 1);
  double *mat = ctx.data() + 2 + nattr;
  for (int k = 0; k < nmat; k++) {
    const double *m = Q->GetMaterialProperty(k);
    if (qd == 1) {
      for (int i = 0; i < dim; i++) mat[(size_t)d2 * k + i * (dim + 1)] = a * m[0];
    } else {
      for (int dj = 0; dj < dim; dj++)
        for (int di = 0; di < dim; di++) mat[(size_t)d2 * k + (transpose ? di * dim + dj : dj * dim + di)] = a * m[di + dim * dj];
    }
  }
  return ctx;
}

std::vector<double> PopulateCoefficientContext(int dim_mass, const MaterialPropertyCoefficient *Q_mass, int dim,
                                               const MaterialPropertyCoefficient *Q, bool transpose_mass, bool transpose,
                                               double a_mass, double a) {
  auto ctx = PopulateCoefficientContext(dim_mass, Q_mass, transpose_mass, a_mass);
  const auto second = PopulateCoefficientContext(dim, Q, transpose, a);
  ctx.insert(ctx.end(), second.begin(), second.end());
  return ctx;
}

}  // namespace ceed

// ---- 1-D bases ---------------------------------------------------------------------------------------------------------
namespace fem {

void GaussLegendre(int n, std::vector<double> &x, std::vector<double> &w) {
  // roots of P_n by Newton from the Chebyshev guess, mapped to [0, 1]
  x.assign((size_t)n, 0.0), w.assign((size_t)n, 0.0);
  for (int i = 0; i < (n + 1) / 2; i++) {
    double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 0.0;
    for (int it = 0; it < 100; it++) {
      double p0 = 1.0, p1 = z;
      for (int k = 2; k <= n; k++) {
        const double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
        p0 = p1, p1 = p2;
      }
      if (n == 1) p0 = 1.0, p1 = z;
      pp = n * (z * p1 - p0) / (z * z - 1.0);
      const double dz = p1 / pp;
      z -= dz;
      if (std::abs(dz) < 1e-16) break;
    }
    {
      double p0 = 1.0, p1 = z;
      for (int k = 2; k <= n; k++) {
        const double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
        p0 = p1, p1 = p2;
      }
      pp = n * (z * p1 - p0) / (z * z - 1.0);
    }
    const double wi = 2.0 / ((1.0 - z * z) * pp * pp);
    x[(size_t)i] = 0.5 * (1.0 - z), x[(size_t)(n - 1 - i)] = 0.5 * (1.0 + z);
    w[(size_t)i] = w[(size_t)(n - 1 - i)] = 0.5 * wi;
  }
  if (n % 2) x[(size_t)(n / 2)] = 0.5;
}

std::vector<double> GaussLobatto(int n) {
  PA_REQUIRE(n >= 2, "Gauss-Lobatto needs >= 2 points");
  std::vector<double> x((size_t)n);
  if (n == 2) return {0.0, 1.0};
  const int N = n - 1;
  for (int i = 0; i < n; i++) {
    double z = -std::cos(M_PI * i / N);
    if (i > 0 && i < N) {
      for (int it = 0; it < 100; it++) {  // Newton on (1 - z^2) P_N'(z)
        double p0 = 1.0, p1 = z;
        for (int k = 2; k <= N; k++) {
          const double p2 = ((2 * k - 1) * z * p1 - (k - 1) * p0) / k;
          p0 = p1, p1 = p2;
        }
        const double dz = (z * p1 - p0) / ((N + 1) * p1);
        z -= dz;
        if (std::abs(dz) < 1e-16) break;
      }
    }
    x[(size_t)i] = z;
  }
  x[0] = -1.0, x[(size_t)N] = 1.0;
  std::vector<double> out((size_t)n);
  for (int i = 0; i < n; i++) out[(size_t)i] = 0.5 * (0.5 * (x[(size_t)i] - x[(size_t)(N - i)]) + 1.0);  // symmetrised
  return out;
}

void LagrangeEval(const std::vector<double> &nodes, const std::vector<double> &x, std::vector<double> &B,
                  std::vector<double> &G) {
  const int n = (int)nodes.size(), nx = (int)x.size();
  B.assign((size_t)nx * n, 0.0), G.assign((size_t)nx * n, 0.0);
  for (int i = 0; i < n; i++) {
    double denom = 1.0;
    for (int m = 0; m < n; m++)
      if (m != i) denom *= nodes[(size_t)i] - nodes[(size_t)m];
    for (int q = 0; q < nx; q++) {
      double val = 1.0, d = 0.0;
      for (int m = 0; m < n; m++)
        if (m != i) val *= x[(size_t)q] - nodes[(size_t)m];
      for (int m = 0; m < n; m++) {
        if (m == i) continue;
        double t = 1.0;
        for (int l = 0; l < n; l++)
          if (l != i && l != m) t *= x[(size_t)q] - nodes[(size_t)l];
        d += t;
      }
      B[(size_t)q * n + i] = val / denom, G[(size_t)q * n + i] = d / denom;
    }
  }
}

}  // namespace fem

// ---- Mesh -----------------------------------------------------------------------------------------------------------
Mesh::Mesh(const Context &ctx, int num_elem, int mesh_order, int num_nodes, const int32_t *node_offsets, const double *nodes,
           const int32_t *attr, int q1d)
    : ne_(num_elem), q1d_(q1d), mesh_order_(mesh_order) {
  // nodal H1 mesh space on equispaced lattice points (MFEM's default nodal basis for the mesh nodes is Gauss-Lobatto;
  // for order 2 both are {0, 1/2, 1})
  PA_REQUIRE(mesh_order == 1 || mesh_order == 2, "mesh order 1 or 2 expected");
  std::vector<double> qx, qw, B, G;
  fem::GaussLegendre(q1d, qx, qw);
  fem::LagrangeEval(fem::GaussLobatto(mesh_order + 1), qx, B, G);
  pa_mesh_desc m{num_elem, mesh_order, q1d, num_nodes, node_offsets, nodes, attr, B.data(), G.data(), qw.data()};
  check(pa_geom_create(&m, ctx.stream, &geom_));
  const int n1 = mesh_order + 1, npe = n1 * n1 * n1;
  ncorner_ = 8;
  corner_nodes_.resize((size_t)num_elem * 8);
  for (int e = 0; e < num_elem; e++)
    for (int c = 0; c < 8; c++) {
      const int i = (c & 1) * mesh_order, j = ((c >> 1) & 1) * mesh_order, k = ((c >> 2) & 1) * mesh_order;
      corner_nodes_[(size_t)e * 8 + c] = node_offsets[(size_t)e * npe + i + n1 * (j + n1 * k)];
    }
  nodes_.assign(nodes, nodes + (size_t)num_nodes * 3);
}
Mesh::Mesh(const Context &ctx, const pa_mesh_dense_desc &desc)
    : ne_(desc.num_elem), q1d_(0), mesh_order_(0), nq_dense_(desc.num_qpts) {
  dim_ = desc.dim == 0 ? 3 : desc.dim;
  sdim_ = desc.space_dim == 0 ? dim_ : desc.space_dim;
  check(pa_geom_create_dense(&desc, ctx.stream, &geom_));
  if (desc.nodes_per_elem >= dim_ + 1) {  // simplices: the vertices are the first dim + 1 nodes (MFEM's node order)
    ncorner_ = dim_ + 1;
    corner_nodes_.resize((size_t)desc.num_elem * ncorner_);
    for (int e = 0; e < desc.num_elem; e++)
      for (int c = 0; c < ncorner_; c++)
        corner_nodes_[(size_t)e * ncorner_ + c] = desc.node_offsets[(size_t)e * desc.nodes_per_elem + c];
    nodes_.assign(desc.nodes, desc.nodes + (size_t)desc.num_nodes * sdim_);
  }
}
std::vector<double> Mesh::VertexCoordinates(const FiniteElementSpace &h1) const {
  PA_REQUIRE(&h1.GetMesh() == this && h1.GetFEType() == PA_FE_H1 && h1.GetMaxElementOrder() == 1 &&
                 h1.GetElemSize() == ncorner_,
             "vertex coordinates: a lowest-order H1 space on this mesh expected");
  std::vector<double> xyz((size_t)h1.GetVSize() * sdim_, 0.0);
  for (int e = 0; e < ne_; e++)
    for (int c = 0; c < ncorner_; c++) {
      const int32_t d = h1.GetElementDof(e, c), nd = corner_nodes_[(size_t)e * ncorner_ + c];
      for (int k = 0; k < sdim_; k++) xyz[(size_t)d * sdim_ + k] = nodes_[(size_t)nd * sdim_ + k];
    }
  return xyz;
}
void Mesh::SetRefinementTransforms(const Mesh &parent, const int32_t *embed_parent, const int32_t *embed_matrix, int nmat,
                                   const double *point_matrices) {
  PA_REQUIRE(embed_parent && embed_matrix && point_matrices && nmat >= 1 && nmat <= 256, "invalid refinement transforms");
  PA_REQUIRE(parent.dim_ == dim_ && parent.ncorner_ == ncorner_ && ncorner_ > 0, "parent and child meshes of one element type expected");
  for (int e = 0; e < ne_; e++) {
    PA_REQUIRE(embed_parent[e] >= 0 && embed_parent[e] < parent.ne_, "refinement: parent element out of range");
    PA_REQUIRE(embed_matrix[e] >= 0 && embed_matrix[e] < nmat, "refinement: point matrix out of range");
  }
  parent_ = &parent;
  embed_parent_.assign(embed_parent, embed_parent + ne_);
  embed_matrix_.assign(embed_matrix, embed_matrix + ne_);
  n_point_matrices_ = nmat;
  point_matrices_.assign(point_matrices, point_matrices + (size_t)nmat * ncorner_ * dim_);
}
Mesh::~Mesh() {
  if (geom_) pa_geom_destroy(geom_);
}

// ---- FiniteElementSpace ------------------------------------------------------------------------------------------------
FiniteElementSpace::FiniteElementSpace(const Context &ctx, const Mesh &mesh, int fe_type, int order, int vsize,
                                       const int32_t *offsets, const uint8_t *orients, const int32_t *dof_map, int n_true,
                                       const Halo *halo)
    : ctx_(&ctx), mesh_(&mesh), fe_type_(fe_type), order_(order), vsize_(vsize), true_vsize_(n_true < 0 ? vsize : n_true),
      halo_(halo) {
  PA_REQUIRE(fe_type == PA_FE_HCURL || fe_type == PA_FE_H1, "H(curl) or H1 space expected");
  PA_REQUIRE(order >= 1 && offsets, "invalid finite element space description");
  elem_size_ = fe_type == PA_FE_HCURL ? 3 * order * (order + 1) * (order + 1) : (order + 1) * (order + 1) * (order + 1);
  const size_t n = (size_t)mesh.GetNE() * elem_size_;
  offsets_.assign(offsets, offsets + n);
  if (orients) orients_.assign(orients, orients + n);
  if (dof_map) dof_map_.assign(dof_map, dof_map + elem_size_);
  std::vector<double> qx, qw, ox, ow, Go;
  fem::GaussLegendre(mesh.GetQ1d(), qx, qw);
  fem::LagrangeEval(fem::GaussLobatto(order + 1), qx, Bc_, Gc_);
  fem::GaussLegendre(order, ox, ow);
  fem::LagrangeEval(ox, qx, Bo_, Go);
}

FiniteElementSpace::FiniteElementSpace(const Context &ctx, const Mesh &mesh, int fe_type, int order, int elem_size, int vsize,
                                       const int32_t *offsets, const uint8_t *orients, const int8_t *curl_orients,
                                       const double *interp, const double *deriv, int n_true, const Halo *halo)
    : ctx_(&ctx), mesh_(&mesh), fe_type_(fe_type), order_(order), elem_size_(elem_size), vsize_(vsize),
      true_vsize_(n_true < 0 ? vsize : n_true), halo_(halo) {
  PA_REQUIRE(mesh.IsDense(), "a space given by dense tables needs a dense Mesh");
  PA_REQUIRE(fe_type == PA_FE_HCURL || fe_type == PA_FE_H1 || fe_type == PA_FE_HDIV, "unknown element type");
  PA_REQUIRE(elem_size > 0 && offsets && (interp || deriv) && !(orients && curl_orients), "invalid finite element space description");
  const size_t n = (size_t)mesh.GetNE() * elem_size, Q = (size_t)mesh.GetNumQuadraturePoints();
  offsets_.assign(offsets, offsets + n);
  if (orients) orients_.assign(orients, orients + n);
  if (curl_orients) curl_orients_.assign(curl_orients, curl_orients + 3 * n);
  // components of the reference-space values and derivatives (basis.cpp:40-85): vectors have dim components, the curl of a
  // 2-D Nedelec element is a scalar
  const size_t dim = (size_t)mesh.Dimension();
  const size_t qcomp = fe_type == PA_FE_H1 ? 1 : dim;
  const size_t dcomp = (fe_type == PA_FE_HDIV || (fe_type == PA_FE_HCURL && dim == 2)) ? 1 : dim;  // divergence, 2-D curl: scalars
  if (interp) interp_.assign(interp, interp + qcomp * Q * elem_size);
  if (deriv) deriv_.assign(deriv, deriv + dcomp * Q * elem_size);
}

pa_restriction_desc FiniteElementSpace::GetCeedElemRestriction() const {
  return pa_restriction_desc{mesh_->GetNE(), elem_size_, vsize_, offsets_.data(), orients_.empty() ? nullptr : orients_.data(),
                             curl_orients_.empty() ? nullptr : curl_orients_.data()};
}
VectorFiniteElementSpace::VectorFiniteElementSpace(const FiniteElementSpace &scalar, int vdim, bool by_vdim)
    : scalar_(&scalar), vdim_(vdim), by_vdim_(by_vdim) {
  PA_REQUIRE(scalar.GetFEType() == PA_FE_H1 && vdim >= 1 && vdim <= 3, "vector spaces: 1-3 copies of an H1 space");
  const pa_restriction_desc r = scalar.GetCeedElemRestriction();
  offsets_.assign(r.offsets, r.offsets + (size_t)r.num_elem * r.elem_size);
  if (by_vdim)
    for (int32_t &o : offsets_) o *= vdim;
}
pa_restriction_desc VectorFiniteElementSpace::GetCeedElemRestriction() const {
  pa_restriction_desc r = scalar_->GetCeedElemRestriction();
  r.offsets = offsets_.data(), r.lsize = GetVSize();
  return r;
}
pa_dense_basis_desc FiniteElementSpace::GetCeedDenseBasis() const {
  PA_REQUIRE(IsDense(), "the space has no dense tables");
  return pa_dense_basis_desc{fe_type_, elem_size_, mesh_->GetNumQuadraturePoints(), interp_.empty() ? nullptr : interp_.data(),
                             deriv_.empty() ? nullptr : deriv_.data()};
}
pa_basis_desc FiniteElementSpace::GetCeedBasis() const {
  return pa_basis_desc{fe_type_, order_, mesh_->GetQ1d(), Bc_.data(), Gc_.data(), Bo_.data(),
                       dof_map_.empty() ? nullptr : dof_map_.data(), nullptr, nullptr};
}

std::pair<int32_t, bool> FiniteElementSpace::GetElementDofSigned(int e, int t) const {
  bool neg = false;
  int j = t;
  if (!dof_map_.empty()) {
    j = dof_map_[t];
    if (j < 0) j = -1 - j, neg = true;
  }
  const size_t k = (size_t)e * elem_size_ + j;
  if (!orients_.empty() && orients_[k]) neg = !neg;
  return {offsets_[k], neg};
}
void FiniteElementSpace::SetLocalInterpolation(const double *M, int nmat) {
  PA_REQUIRE(IsDense() && M && nmat == mesh_->GetNumPointMatrices(), "local interpolation: one matrix per point matrix of the dense mesh");
  local_interp_.assign(M, M + (size_t)nmat * elem_size_ * elem_size_);
}

const Operator &FiniteElementSpace::GetDiscreteInterpolator(const FiniteElementSpace &aux) const {
  auto it = G_.find(&aux);
  if (it != G_.end()) return *it->second;
  PA_REQUIRE(fe_type_ == PA_FE_HCURL && aux.fe_type_ == PA_FE_H1 && aux.order_ == order_,
             "the discrete gradient maps the H1 space of the same order into the Nedelec space");
  // derivative of the closed basis at the open nodes, [p][p + 1]; the identity in the other directions
  std::vector<double> ox, ow, Bg, Dg;
  fem::GaussLegendre(order_, ox, ow);
  fem::LagrangeEval(fem::GaussLobatto(order_ + 1), ox, Bg, Dg);
  const int n = order_ + 1;
  std::vector<double> I((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
  const auto rh = aux.GetCeedElemRestriction(), rn = GetCeedElemRestriction();
  const auto bh = aux.GetCeedBasis(), bn = GetCeedBasis();
  auto &slot = G_[&aux];
  slot.reset(make_interp_operator(*ctx_, rh, bh, rn, bn, I.data(), Dg.data(), aux.halo_, aux.true_vsize_, true_vsize_, 1));
  return *slot;
}

const Operator &FiniteElementSpaceHierarchy::BuildProlongationAtLevel(std::size_t l) const {
  // p-prolongation on one mesh (fespace.cpp:188-203 + bilinearform.cpp:203-282): Kronecker products of the 1-D nodal
  // interpolation matrices between the closed / open point sets of the two orders
  const FiniteElementSpace &c = *fespaces_.at(l), &f = *fespaces_.at(l + 1);
  if (&c.GetMesh() != &f.GetMesh()) {
    // two meshes: the refinement transfer (fespace.cpp:246-251, mfem::TransferOperator): over the FINE elements, the parent's
    // dofs -> the child's dofs through the local interpolation matrix of the child's embedding
    const Mesh &mf = f.GetMesh();
    PA_REQUIRE(mf.GetParent() == &c.GetMesh(), "levels on different meshes: the finer mesh must be a refinement of the coarser one "
                                               "(Mesh::SetRefinementTransforms)");
    PA_REQUIRE(c.GetFEType() == f.GetFEType() && c.GetMaxElementOrder() == f.GetMaxElementOrder() && c.GetElemSize() == f.GetElemSize(),
               "h-levels carry the same finite element collection on every mesh (fem/multigrid.hpp:103-112)");
    PA_REQUIRE(c.IsDense() == f.IsDense() && !c.GetHalo() && !f.GetHalo(), "refinement transfer: one rank, one kind of space");
    const int P = f.GetElemSize(), nef = mf.GetNE(), nmat = mf.GetNumPointMatrices(), p = f.GetMaxElementOrder();
    std::vector<double> M;
    if (f.IsDense()) {
      M = f.GetLocalInterpolation();
      PA_REQUIRE(M.size() == (size_t)nmat * P * P, "dense spaces on a refined mesh need FiniteElementSpace::SetLocalInterpolation");
    } else {
      // tensor elements: child = an axis-aligned box o + s x of the parent's reference cube (corners in lexicographic order);
      // closed directions: parent basis at the child's nodes; open direction (H(curl)): the same times the tangent's scale s
      PA_REQUIRE(mf.GetNumCorners() == 8 && mf.Dimension() == 3, "tensor blocks of hexahedra expected");
      const bool hcurl = f.GetFEType() == PA_FE_HCURL;
      const std::vector<double> cp = fem::GaussLobatto(p + 1);
      std::vector<double> op, ow;
      if (hcurl) fem::GaussLegendre(p, op, ow);
      M.assign((size_t)nmat * P * P, 0.0);
      for (int m = 0; m < nmat; m++) {
        const double *pm = mf.GetPointMatrices().data() + (size_t)m * 8 * 3;
        double o[3], sc[3];
        for (int d = 0; d < 3; d++) o[d] = pm[d], sc[d] = pm[3 * (1 << d) + d] - pm[d];
        for (int v = 0; v < 8; v++)
          for (int d = 0; d < 3; d++)
            PA_REQUIRE(std::fabs(pm[3 * v + d] - (o[d] + ((v >> d) & 1) * sc[d])) < 1e-12 && sc[d] > 0.0,
                       "refinement: the child is not an axis-aligned box of its parent's reference cube");
        // 1-D matrices [child node][parent function] per direction, closed and open
        std::vector<double> Ic[3], Io[3], tmp, x;
        for (int d = 0; d < 3; d++) {
          x.resize(cp.size());
          for (size_t i = 0; i < cp.size(); i++) x[i] = o[d] + sc[d] * cp[i];
          fem::LagrangeEval(cp, x, Ic[d], tmp);
          if (hcurl) {
            x.resize(op.size());
            for (size_t i = 0; i < op.size(); i++) x[i] = o[d] + sc[d] * op[i];
            fem::LagrangeEval(op, x, Io[d], tmp);
            for (double &v : Io[d]) v *= sc[d];
          }
        }
        double *Mm = M.data() + (size_t)m * P * P;
        const int n1 = p + 1, ncomp = hcurl ? 3 : 1, nblk = hcurl ? p * n1 * n1 : n1 * n1 * n1;
        for (int comp = 0; comp < ncomp; comp++) {
          int nd[3] = {n1, n1, n1};
          const std::vector<double> *I1[3] = {&Ic[0], &Ic[1], &Ic[2]};
          if (hcurl) nd[comp] = p, I1[comp] = &Io[comp];
          for (int kf = 0; kf < nd[2]; kf++)
            for (int jf = 0; jf < nd[1]; jf++)
              for (int i_f = 0; i_f < nd[0]; i_f++)
                for (int kc = 0; kc < nd[2]; kc++)
                  for (int jc = 0; jc < nd[1]; jc++)
                    for (int ic = 0; ic < nd[0]; ic++) {
                      const int row = comp * nblk + i_f + nd[0] * (jf + nd[1] * kf), col = comp * nblk + ic + nd[0] * (jc + nd[1] * kc);
                      Mm[(size_t)row * P + col] = (*I1[0])[(size_t)i_f * nd[0] + ic] * (*I1[1])[(size_t)jf * nd[1] + jc] *
                                                  (*I1[2])[(size_t)kf * nd[2] + kc];
                    }
        }
      }
    }
    // restrictions in the order of the matrices (tensor order for tensor spaces, native for dense ones)
    std::vector<int32_t> off_d((size_t)nef * P), off_r((size_t)nef * P);
    std::vector<uint8_t> ori_d((size_t)nef * P), ori_r((size_t)nef * P), mid((size_t)nef);
    bool any_sign = false;
    for (int e = 0; e < nef; e++) {
      const int E = mf.GetEmbeddingParents()[e];
      mid[e] = (uint8_t)mf.GetEmbeddingMatrices()[e];
      for (int t = 0; t < P; t++) {
        const auto dc = c.GetElementDofSigned(E, t), df = f.GetElementDofSigned(e, t);
        off_d[(size_t)e * P + t] = dc.first, ori_d[(size_t)e * P + t] = dc.second;
        off_r[(size_t)e * P + t] = df.first, ori_r[(size_t)e * P + t] = df.second;
        any_sign = any_sign || dc.second || df.second;
      }
    }
    const pa_restriction_desc rd{nef, P, c.GetVSize(), off_d.data(), any_sign ? ori_d.data() : nullptr, nullptr};
    const pa_restriction_desc rr{nef, P, f.GetVSize(), off_r.data(), any_sign ? ori_r.data() : nullptr, nullptr};
    P_[l].reset(make_dense_interp_operator(c.GetContext(), rd, rr, M.data(), nullptr, c.GetTrueVSize(), f.GetTrueVSize(), nmat,
                                           mid.data()));
    return *P_[l];
  }
  std::vector<double> Ic, Io, tmp, xc, wc, xf, wf;
  fem::LagrangeEval(fem::GaussLobatto(c.GetMaxElementOrder() + 1), fem::GaussLobatto(f.GetMaxElementOrder() + 1), Ic, tmp);
  fem::GaussLegendre(c.GetMaxElementOrder(), xc, wc);
  fem::GaussLegendre(f.GetMaxElementOrder(), xf, wf);
  fem::LagrangeEval(xc, xf, Io, tmp);
  const auto rc = c.GetCeedElemRestriction(), rf = f.GetCeedElemRestriction();
  const auto bc = c.GetCeedBasis(), bf = f.GetCeedBasis();
  P_[l].reset(make_interp_operator(c.GetContext(), rc, bc, rf, bf, Ic.data(), Io.data(), c.GetHalo(), c.GetTrueVSize(),
                                   f.GetTrueVSize(), 0));
  return *P_[l];
}

std::vector<const Operator *> FiniteElementSpaceHierarchy::GetProlongationOperators() const {
  PA_REQUIRE(GetNumLevels() > 1, "Out of bounds request for finite element space prolongation at level 0!");
  std::vector<const Operator *> P(GetNumLevels() - 1);
  for (std::size_t l = 0; l < P.size(); l++) P[l] = &GetProlongationAtLevel(l);
  return P;
}

std::vector<const Operator *> FiniteElementSpaceHierarchy::GetDiscreteInterpolators(
    const FiniteElementSpaceHierarchy &aux_fespaces) const {
  std::vector<const Operator *> G(GetNumLevels(), nullptr);
  for (std::size_t l = 1; l < G.size(); l++) G[l] = &GetFESpaceAtLevel(l).GetDiscreteInterpolator(aux_fespaces.GetFESpaceAtLevel(l));
  return G;
}

// ---- integrators ----------------------------------------------------------------------------------------------------
void BilinearFormIntegrator::AssembleCeedOperator(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test,
                                                  int qf, const std::vector<double> &ctx, int trial_ops, int test_ops) {
  if (trial.IsDense() || test.IsDense()) {
    PA_REQUIRE(trial.IsDense() && test.IsDense() && &trial.GetMesh() == &test.GetMesh(), "dense spaces on one dense mesh expected");
    const auto r = trial.GetCeedElemRestriction();
    const auto b = trial.GetCeedDenseBasis();
    if (&trial == &test) {
      check(pa_op_add_sub_dense(op, trial.GetMesh().GetCeedGeomFactorData(), &r, &b, qf, ctx.data(), ctx.size() * sizeof(double),
                                trial_ops, test_ops));
    } else {  // two spaces: values of vector elements, gradients of H1 elements (pa_op_add_sub_dense_mixed)
      const bool scalar = qf == PA_QF_H1_1;  // MassIntegrator between two scalar spaces: values on both sides
      const bool curls = qf == PA_QF_HDIV_33;  // ... f_apply_hdiv_33 between two spaces: the curls of an H(curl) side
      auto want = [&](const FiniteElementSpace &fes) {
        if (curls && fes.GetFEType() == PA_FE_HCURL) return (int)PA_EVAL_CURL;
        return (int)((fes.GetFEType() == PA_FE_H1 && !scalar) ? PA_EVAL_GRAD : PA_EVAL_INTERP);
      };
      PA_REQUIRE(trial_ops == want(trial) && test_ops == want(test),
                 "mixed-space forms evaluate the values of vector or scalar elements, the gradients of H1 elements and, with "
                 "f_apply_hdiv_33, the curls of H(curl) elements");
      const auto r2 = test.GetCeedElemRestriction();
      const auto b2 = test.GetCeedDenseBasis();
      check(pa_op_add_sub_dense_mixed(op, trial.GetMesh().GetCeedGeomFactorData(), &r, &b, &r2, &b2, qf, ctx.data(),
                                      ctx.size() * sizeof(double)));
    }
    return;
  }
  PA_REQUIRE(&trial == &test, "square forms only: test and trial space must be the same object");
  const auto r = trial.GetCeedElemRestriction();
  const auto b = trial.GetCeedBasis();
  check(pa_op_add_sub(op, trial.GetMesh().GetCeedGeomFactorData(), &r, &b, qf, ctx.data(), ctx.size() * sizeof(double),
                      trial_ops, test_ops));
}

namespace {
// 10 * space_dim + dim of the element block, as the reference's integrators switch on it
int dims_of(const FiniteElementSpace &fes) { return 10 * fes.GetMesh().SpaceDimension() + fes.GetMesh().Dimension(); }
}  // namespace

void MassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  AssembleCeedOperator(op, trial, test, PA_QF_H1_1, ceed::PopulateCoefficientContext(1, Q, transpose), PA_EVAL_INTERP,
                       PA_EVAL_INTERP);
}
void VectorFEMassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // the QFunction follows the map types of the two elements (vecfemass.cpp:75-101)
  const bool tc = trial.GetFEType() == PA_FE_HCURL, sc = test.GetFEType() == PA_FE_HCURL;
  PA_REQUIRE((tc || trial.GetFEType() == PA_FE_HDIV) && (sc || test.GetFEType() == PA_FE_HDIV),
             "Invalid trial/test element map type for VectorFEMassIntegrator!");
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31, "Invalid value of (dim, space_dim) for VectorFEMassIntegrator!");
  // {H(curl) x H(curl), H(curl) trial x H(div) test, H(div) trial x H(curl) test, H(div) x H(div)} of every geometry
  static const int table[5][4] = {{PA_QF_HCURL_33, PA_QF_HCURLHDIV_33, PA_QF_HDIVHCURL_33, PA_QF_HDIV_33},
                                  {PA_QF_HCURL_22, PA_QF_HCURLHDIV_22, PA_QF_HDIVHCURL_22, PA_QF_HDIV_22},
                                  {PA_QF_HCURL_32, PA_QF_HCURLHDIV_32, PA_QF_HDIVHCURL_32, PA_QF_HDIV_32},
                                  {PA_QF_HCURL_21, PA_QF_HCURLHDIV_21, PA_QF_HDIVHCURL_21, PA_QF_HDIV_21},
                                  {PA_QF_HCURL_31, PA_QF_HCURLHDIV_31, PA_QF_HDIVHCURL_31, PA_QF_HDIV_31}};
  const int qf = table[d == 33 ? 0 : d == 22 ? 1 : d == 32 ? 2 : d == 21 ? 3 : 4][tc ? (sc ? 0 : 1) : (sc ? 2 : 3)];
  AssembleCeedOperator(op, trial, test, qf, ceed::PopulateCoefficientContext(sdim, Q, transpose), PA_EVAL_INTERP, PA_EVAL_INTERP);
}
void DiffusionIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31, "Invalid value of (dim, space_dim) for DiffusionIntegrator!");
  const int qf_diff = d == 33 ? PA_QF_HCURL_33
                      : (d == 22 ? PA_QF_HCURL_22 : (d == 32 ? PA_QF_HCURL_32 : (d == 21 ? PA_QF_HCURL_21 : PA_QF_HCURL_31)));
  AssembleCeedOperator(op, trial, test, qf_diff,
                       ceed::PopulateCoefficientContext(sdim, Q, transpose), PA_EVAL_GRAD, PA_EVAL_GRAD);
}
void CurlCurlIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  const int d = dims_of(trial), dim = trial.GetMesh().Dimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32, "Invalid value of (dim, space_dim) for CurlCurlIntegrator!");
  // the curl of a 2-D element has a single component: scalar coefficient, Weight input on the trial side (curlcurl.cpp:40-72)
  AssembleCeedOperator(op, trial, test, d == 33 ? PA_QF_HDIV_33 : PA_QF_L2_1,
                       ceed::PopulateCoefficientContext(dim < 3 ? 1 : dim, Q, transpose),
                       PA_EVAL_CURL | (dim < 3 ? PA_EVAL_WEIGHT : 0), PA_EVAL_CURL);
}
void DivDivIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  PA_REQUIRE(trial.GetFEType() == PA_FE_HDIV && test.GetFEType() == PA_FE_HDIV, "DivDivIntegrator: H(div) spaces expected");
  AssembleCeedOperator(op, trial, test, PA_QF_L2_1, ceed::PopulateCoefficientContext(1, Q, transpose),
                       PA_EVAL_DIV | PA_EVAL_WEIGHT, PA_EVAL_DIV);  // divdiv.cpp:31-57 (single-component elements)
}
void MixedVectorGradientIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // mixedvecgrad.cpp:43-76: the QFunction follows the map type of the test element; trial Grad, test Interp (:117-118)
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  const bool sc = test.GetFEType() == PA_FE_HCURL;
  PA_REQUIRE(trial.GetFEType() == PA_FE_H1 && (sc || test.GetFEType() == PA_FE_HDIV),
             "Invalid trial/test element map type for MixedVectorGradientIntegrator!");
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31,
             "Invalid value of (dim, space_dim) for MixedVectorGradientIntegrator!");
  const int qf = d == 33   ? (sc ? PA_QF_HCURL_33 : PA_QF_HCURLHDIV_33)
                 : d == 22 ? (sc ? PA_QF_HCURL_22 : PA_QF_HCURLHDIV_22)
                 : d == 32 ? (sc ? PA_QF_HCURL_32 : PA_QF_HCURLHDIV_32)
                 : d == 21 ? (sc ? PA_QF_HCURL_21 : PA_QF_HCURLHDIV_21)
                           : (sc ? PA_QF_HCURL_31 : PA_QF_HCURLHDIV_31);
  AssembleCeedOperator(op, trial, test, qf, ceed::PopulateCoefficientContext(sdim, Q, transpose), PA_EVAL_GRAD, PA_EVAL_INTERP);
}
void MixedVectorWeakDivergenceIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // mixedvecgrad.cpp:146-202: f_apply_hcurl_* of the geometry, trial Interp, test Grad, the coefficient scaled by -1
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(trial.GetFEType() == PA_FE_HCURL && test.GetFEType() == PA_FE_H1,
             "MixedVectorWeakDivergenceIntegrator: H(curl) trial space, H1 test space");
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31,
             "Invalid value of (dim, space_dim) for MixedVectorWeakDivergenceIntegrator!");
  const int qf = d == 33 ? PA_QF_HCURL_33 : (d == 22 ? PA_QF_HCURL_22 : (d == 32 ? PA_QF_HCURL_32 : (d == 21 ? PA_QF_HCURL_21 : PA_QF_HCURL_31)));
  AssembleCeedOperator(op, trial, test, qf, ceed::PopulateCoefficientContext(sdim, Q, transpose, -1.0), PA_EVAL_INTERP, PA_EVAL_GRAD);
}
void VectorMassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  AssembleCeedOperator(op, trial, test, PA_QF_H1_1, ceed::PopulateCoefficientContext(1, Q, transpose), PA_EVAL_INTERP, PA_EVAL_INTERP);
}
void VectorMassIntegrator::Assemble(pa_op *op, const VectorFiniteElementSpace &fes) const {
  const FiniteElementSpace &comp = fes.GetScalarSpace();
  PA_REQUIRE(comp.IsDense() && fes.GetVDim() >= 2, "MassIntegrator with several components: a dense-table H1 space, 2 or 3 copies");
  const std::vector<double> ctx = ceed::PopulateCoefficientContext(fes.GetVDim(), Q, transpose);
  const auto r = fes.GetCeedElemRestriction();
  const auto b = comp.GetCeedDenseBasis();
  check(pa_op_add_sub_dense_vector_mass(op, comp.GetMesh().GetCeedGeomFactorData(), &r, &b, fes.GetVDim(), fes.GetCompStride(),
                                        ctx.data(), ctx.size() * sizeof(double)));
}
std::unique_ptr<ceed::Operator> VectorMassIntegrator::PartialAssemble(const VectorFiniteElementSpace &fes) const {
  pa_op *op = nullptr;
  check(pa_op_create(fes.GetVSize(), fes.GetVSize(), &op));
  auto out = std::make_unique<ceed::Operator>(fes.GetScalarSpace().GetContext(), op, /*own=*/true);
  Assemble(op, fes);
  check(pa_op_finalize(op));
  return out;
}
void GradientIntegrator::Assemble(pa_op *, const FiniteElementSpace &, const FiniteElementSpace &) const {
  throw pa::Error("GradientIntegrator requires trial space with a single component and test space with space_dim components! "
                  "(Assemble(op, trial, VectorFiniteElementSpace))");
}
void GradientIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const VectorFiniteElementSpace &test) const {
  const FiniteElementSpace &comp = test.GetScalarSpace();
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(trial.GetFEType() == PA_FE_H1 && test.GetVDim() == sdim && trial.IsDense() && comp.IsDense() &&
                 &trial.GetMesh() == &comp.GetMesh(),
             "GradientIntegrator requires trial space with a single component and test space with space_dim components!");
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31, "Invalid value of (dim, space_dim) for GradientIntegrator!");
  const int qf = d == 33 ? PA_QF_HCURLH1D_33
                 : (d == 22 ? PA_QF_HCURLH1D_22 : (d == 32 ? PA_QF_HCURLH1D_32 : (d == 21 ? PA_QF_HCURLH1D_21 : PA_QF_HCURLH1D_31)));
  const std::vector<double> ctx = ceed::PopulateCoefficientContext(sdim, Q, transpose);
  const auto r1 = trial.GetCeedElemRestriction(), r2 = test.GetCeedElemRestriction();
  const auto b1 = trial.GetCeedDenseBasis(), b2 = comp.GetCeedDenseBasis();
  check(pa_op_add_sub_dense_gradient(op, trial.GetMesh().GetCeedGeomFactorData(), &r1, &b1, &r2, &b2, test.GetCompStride(), qf,
                                     ctx.data(), ctx.size() * sizeof(double)));
}
std::unique_ptr<ceed::Operator> GradientIntegrator::PartialAssemble(const FiniteElementSpace &trial,
                                                                    const VectorFiniteElementSpace &test) const {
  pa_op *op = nullptr;
  check(pa_op_create(test.GetVSize(), trial.GetVSize(), &op));
  auto out = std::make_unique<ceed::Operator>(trial.GetContext(), op, /*own=*/true);
  Assemble(op, trial, test);
  check(pa_op_finalize(op));
  return out;
}
void MixedVectorCurlIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // mixedveccurl.cpp:21-73: (Q curl u, v); the QFunction follows the map type of the test element
  const bool sc = test.GetFEType() == PA_FE_HCURL;
  PA_REQUIRE(trial.GetFEType() == PA_FE_HCURL && (sc || test.GetFEType() == PA_FE_HDIV),
             "Invalid trial/test element map type for MixedVectorCurlIntegrator!");
  PA_REQUIRE(dims_of(trial) == 33, "MixedVectorCurlIntegrator is only available in 3D!");
  AssembleCeedOperator(op, trial, test, sc ? PA_QF_HDIVHCURL_33 : PA_QF_HDIV_33, ceed::PopulateCoefficientContext(3, Q, transpose),
                       PA_EVAL_CURL, PA_EVAL_INTERP);
}
void MixedVectorWeakCurlIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // mixedveccurl.cpp:75-120: -(Q u, curl v) -- the coefficient enters scaled by -1 (:111); the QFunction follows the map type of
  // the trial element
  const bool tc = trial.GetFEType() == PA_FE_HCURL;
  PA_REQUIRE(test.GetFEType() == PA_FE_HCURL && (tc || trial.GetFEType() == PA_FE_HDIV),
             "Invalid trial/test element map type for MixedVectorWeakCurlIntegrator!");
  PA_REQUIRE(dims_of(trial) == 33, "MixedVectorWeakCurlIntegrator is only available in 3D!");
  AssembleCeedOperator(op, trial, test, tc ? PA_QF_HCURLHDIV_33 : PA_QF_HDIV_33,
                       ceed::PopulateCoefficientContext(3, Q, transpose, -1.0), PA_EVAL_INTERP, PA_EVAL_CURL);
}
void DiffusionMassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31, "Invalid value of (dim, space_dim) for DiffusionMassIntegrator!");
  const int qf_dm = d == 33 ? PA_QF_HCURLMASS_33
                    : (d == 22 ? PA_QF_HCURLMASS_22
                               : (d == 32 ? PA_QF_HCURLMASS_32 : (d == 21 ? PA_QF_HCURLMASS_21 : PA_QF_HCURLMASS_31)));
  AssembleCeedOperator(op, trial, test, qf_dm,
                       ceed::PopulateCoefficientContext(1, Q_mass, sdim, Q, transpose_mass, transpose),
                       PA_EVAL_GRAD | PA_EVAL_INTERP, PA_EVAL_GRAD | PA_EVAL_INTERP);
}
void DivDivMassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  // divdivmass.cpp:16-78: f_apply_l2mass_* of the geometry; Interp | Div | Weight on the trial side, mass context first
  PA_REQUIRE(trial.GetFEType() == PA_FE_HDIV && test.GetFEType() == PA_FE_HDIV, "DivDivMassIntegrator: H(div) spaces expected");
  const int d = dims_of(trial), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32 || d == 21 || d == 31, "Invalid value of (dim, space_dim) for DivDivMassIntegrator!");
  const int qf = d == 33 ? PA_QF_L2MASS_33
                 : (d == 22 ? PA_QF_L2MASS_22 : (d == 32 ? PA_QF_L2MASS_32 : (d == 21 ? PA_QF_L2MASS_21 : PA_QF_L2MASS_31)));
  AssembleCeedOperator(op, trial, test, qf, ceed::PopulateCoefficientContext(sdim, Q_mass, 1, Q, transpose_mass, transpose),
                       PA_EVAL_DIV | PA_EVAL_INTERP | PA_EVAL_WEIGHT, PA_EVAL_DIV | PA_EVAL_INTERP);
}
void CurlCurlMassIntegrator::Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const {
  const int d = dims_of(trial), dim = trial.GetMesh().Dimension(), sdim = trial.GetMesh().SpaceDimension();
  PA_REQUIRE(d == 33 || d == 22 || d == 32, "Invalid value of (dim, space_dim) for CurlCurlMassIntegrator!");
  AssembleCeedOperator(op, trial, test, d == 33 ? PA_QF_HDIVMASS_33 : (d == 22 ? PA_QF_HDIVMASS_22 : PA_QF_HDIVMASS_32),
                       ceed::PopulateCoefficientContext(sdim, Q_mass, dim < 3 ? 1 : dim, Q, transpose_mass, transpose),
                       PA_EVAL_CURL | PA_EVAL_INTERP | (dim < 3 ? PA_EVAL_WEIGHT : 0), PA_EVAL_CURL | PA_EVAL_INTERP);
}

// ---- BilinearForm (bilinearform.cpp:27-201) -------------------------------------------------------------------------
std::unique_ptr<ceed::Operator> BilinearForm::PartialAssemble(const FiniteElementSpace &trial,
                                                              const FiniteElementSpace &test) const {
  pa_op *op = nullptr;
  check(pa_op_create(test.GetVSize(), trial.GetVSize(), &op));
  auto out = std::make_unique<ceed::Operator>(trial.GetContext(), op, /*own=*/true);
  for (const auto &integ : domain_integs) integ->Assemble(op, trial, test);
  for (const auto &[bfes, integ] : boundary_integs) {
    PA_REQUIRE(&trial == &test && bfes->GetVSize() == trial.GetVSize() &&
                   bfes->GetMesh().Dimension() == trial.GetMesh().Dimension() - 1,
               "a boundary integrator needs the boundary-element view of the form's (square) space");
    integ->Assemble(op, *bfes, *bfes);
  }
  check(pa_op_finalize(op));
  return out;
}

std::unique_ptr<CsrMatrix> BilinearForm::FullAssemble(const ceed::Operator &op, bool skip_zeros) {
  pa_csr *m = nullptr;
  check(pa_op_full_assemble(op.Handle(), skip_zeros ? 1 : 0, op.GetContext().stream, &m));
  return std::make_unique<CsrMatrix>(op.GetContext(), m);
}

std::unique_ptr<Operator> BilinearForm::Assemble(bool skip_zeros) const {
  if (trial_fespace.GetMaxElementOrder() < pa_order_threshold) return FullAssemble(skip_zeros);
  return PartialAssemble();
}

std::vector<std::unique_ptr<Operator>> BilinearForm::Assemble(const FiniteElementSpaceHierarchy &fespaces, bool skip_zeros,
                                                              std::size_t l0) const {
  PhaseRange range("Operator Construction");  // the drivers hold Timer::CONSTRUCT around SpaceOperator assembly
  PA_REQUIRE(&trial_fespace == &test_fespace && &fespaces.GetFinestFESpace() == &trial_fespace,
             "Assembly on a FiniteElementSpaceHierarchy should have the same BilinearForm spaces and fine space of the "
             "hierarchy!");
  PA_REQUIRE(l0 < fespaces.GetNumLevels(), "No levels available for operator coarsening!");
  PA_REQUIRE(boundary_integs.empty(), "forms with boundary integrators are assembled level by level (one boundary view per level)");
  std::vector<std::unique_ptr<ceed::Operator>> pa_ops;
  for (std::size_t l = l0; l < fespaces.GetNumLevels(); l++) {
    if (l > l0 && &fespaces.GetFESpaceAtLevel(l).GetMesh() == &fespaces.GetFESpaceAtLevel(l - 1).GetMesh())
      pa_ops.push_back(ceed::CeedOperatorCoarsen(*pa_ops.back(), fespaces.GetFESpaceAtLevel(l)));
    else
      pa_ops.push_back(PartialAssemble(fespaces.GetFESpaceAtLevel(l), fespaces.GetFESpaceAtLevel(l)));
  }
  std::vector<std::unique_ptr<Operator>> ops;
  for (std::size_t l = l0; l < fespaces.GetNumLevels(); l++) {
    if (fespaces.GetFESpaceAtLevel(l).GetMaxElementOrder() < pa_order_threshold)
      ops.push_back(FullAssemble(*pa_ops[l - l0], skip_zeros));
    else
      ops.push_back(std::move(pa_ops[l - l0]));
  }
  return ops;
}

namespace ceed {
std::unique_ptr<Operator> CeedOperatorCoarsen(const Operator &op_fine, const FiniteElementSpace &fespace_coarse) {
  const auto r = fespace_coarse.GetCeedElemRestriction();
  const auto b = fespace_coarse.GetCeedBasis();
  pa_op *op = nullptr;
  check(pa_op_coarsen(op_fine.Handle(), &r, &b, &op));
  return std::make_unique<Operator>(fespace_coarse.GetContext(), op, /*own=*/true);
}
}  // namespace ceed

// ---- FespaceParOperator ---------------------------------------------------------------------------------------------
FespaceParOperator::FespaceParOperator(std::unique_ptr<Operator> &&A, const FiniteElementSpace &fespace)
    : Operator(fespace.GetTrueVSize(), fespace.GetTrueVSize()), ctx_(&fespace.GetContext()), local_(std::move(A)),
      fespace_(&fespace) {
  PA_REQUIRE(local_ && local_->Height() == fespace.GetVSize() && local_->Width() == fespace.GetVSize(),
             "the local operator does not match the finite element space");
  par_ = std::make_unique<ParOperator>(*ctx_, *local_, fespace.GetTrueVSize(), nullptr, 0, ParOperator::DiagonalPolicy::DIAG_ONE,
                                       fespace.GetHalo());
}

void FespaceParOperator::SetEssentialTrueDofs(const std::vector<int32_t> &tdofs, ParOperator::DiagonalPolicy policy) {
  ess_tdofs_ = tdofs;
  par_.reset();  // (releases the fused essential list of the local operator before a new wrapper claims it)
  par_ = std::make_unique<ParOperator>(*ctx_, *local_, fespace_->GetTrueVSize(), tdofs.data(), (int)tdofs.size(), policy,
                                       fespace_->GetHalo());
}

}  // namespace palace
