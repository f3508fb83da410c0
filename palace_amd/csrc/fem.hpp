// Host-side C++ mirror of the palace::fem front end of the hot path: the classes a Palace driver touches between
// "here is a mesh and a finite element space" and "here is the operator / the multigrid hierarchy", written against
// this library's C ABI instead of libCEED.  MFEM is not here: what Palace gets from MFEM (element -> dof tables with
// orientations, node coordinates, attributes) comes in as plain arrays, exactly the arrays fem/libceed/restriction.cpp
// and fem/mesh.cpp:146-209 hand to libCEED today; everything else (1-D bases, quadrature, geometry factors, q-data,
// interpolators) is built here.
//
//   MaterialPropertyCoefficient            models/materialoperator.hpp:178-217, materialoperator.cpp:586-868
//   ceed::PopulateCoefficientContext       fem/libceed/coefficient.cpp:51-131
//   fem::DefaultIntegrationOrder           fem/integrator.hpp:25-36, integrator.cpp:14-22
//   Mesh (geometry factor data)            fem/mesh.hpp:27-69, fem/mesh.cpp:146-209
//   FiniteElementSpace / ...Hierarchy      fem/fespace.hpp:22-286, fespace.cpp:27-203
//   BilinearFormIntegrator and derived     fem/integrator.hpp:39-196, fem/integ/{curlcurl,vecfemass,curlcurlmass,
//                                          diffusion,mass,diffusionmass}.cpp
//   BilinearForm                           fem/bilinearform.hpp:21-111, bilinearform.cpp:27-201
//   MultigridOperator                      linalg/operator.hpp:424-493
// Scope: tensor-product hexahedra (the sum-factorised kernels); tetrahedra go through the dense-table entry points of
// the C ABI directly (include/palace_amd.h, pa_op_add_sub_dense).
#pragma once

#include <array>
#include <map>
#include <memory>
#include <vector>

#include "comm.hpp"
#include "linalg.hpp"

namespace palace {

// ---- coefficients -------------------------------------------------------------------------------------------------
// Piecewise-constant matrix coefficient: attribute (1-based) -> material index -> dim x dim (or 1 x 1) matrix.
class MaterialPropertyCoefficient {
  std::vector<int> attr_mat_;      // [attr_max], -1 = no material (zero coefficient)
  int dim_ = 0;                    // rows = columns of every matrix (0 until the first one is added)
  std::vector<double> mat_coeff_;  // [num_mat][dim * dim], column-major like mfem::DenseMatrix
  int num_mat_ = 0;
  double *Mat(int k) { return mat_coeff_.data() + (size_t)k * dim_ * dim_; }
  const double *Mat(int k) const { return mat_coeff_.data() + (size_t)k * dim_ * dim_; }
  void Resize(int dim, int num_mat);
  void UpdateProperty(int k, const double *coeff, int cdim, double a);
  bool Equals(int k, const double *coeff, int cdim, double a) const;

public:
  explicit MaterialPropertyCoefficient(int attr_max);
  MaterialPropertyCoefficient(const std::vector<int> &attr_mat, int dim, const std::vector<double> &mat_coeff, double a = 1.0);
  bool empty() const { return num_mat_ == 0 || dim_ == 0; }
  const std::vector<int> &GetAttributeToMaterial() const { return attr_mat_; }
  int NumMaterials() const { return num_mat_; }
  int Dimension() const { return dim_; }
  const double *GetMaterialProperty(int k) const { return Mat(k); }
  void AddCoefficient(const std::vector<int> &attr_mat, int dim, const std::vector<double> &mat_coeff, double a = 1.0);
  // matrix coefficient (dim x dim, column-major) or scalar on a list of attributes
  void AddMaterialProperty(const std::vector<int> &attr_list, const double *coeff, int dim, double a = 1.0);
  void AddMaterialProperty(const std::vector<int> &attr_list, double coeff, double a = 1.0) {
    AddMaterialProperty(attr_list, &coeff, 1, a);
  }
  void AddMaterialProperty(int attr, double coeff, double a = 1.0) { AddMaterialProperty(std::vector<int>{attr}, coeff, a); }
  MaterialPropertyCoefficient &operator*=(double a);
  void RestrictCoefficient(const std::vector<int> &attr_list);
  void NormalProjectedCoefficient(const std::array<double, 3> &normal);
};

namespace ceed {
// The QFunction context blob (8-byte slots: int in the low half or a double): {num_attr, attr -> mat..., num_mat, matrices}
std::vector<double> PopulateCoefficientContext(int dim, const MaterialPropertyCoefficient *Q, bool transpose = false,
                                               double a = 1.0);
// the paired context of the combined integrators: mass first, then the second-order term (coefficient.cpp:120-131)
std::vector<double> PopulateCoefficientContext(int dim_mass, const MaterialPropertyCoefficient *Q_mass, int dim,
                                               const MaterialPropertyCoefficient *Q, bool transpose_mass = false,
                                               bool transpose = false, double a_mass = 1.0, double a = 1.0);
}  // namespace ceed

namespace fem {

// 1-D nodal bases of Palace's collections (fem/multigrid.hpp:35,49: closed = Gauss-Lobatto, open = Gauss-Legendre) on [0,1]
void GaussLegendre(int n, std::vector<double> &x, std::vector<double> &w);
std::vector<double> GaussLobatto(int n);
// B[q * n + i] = l_i(x_q), G[q * n + i] = l_i'(x_q) for the Lagrange basis on `nodes`
void LagrangeEval(const std::vector<double> &nodes, const std::vector<double> &x, std::vector<double> &B,
                  std::vector<double> &G);

struct DefaultIntegrationOrder {
  inline static int p_trial = 1;
  inline static bool q_order_jac = false;
  inline static int q_order_extra_pk = 0, q_order_extra_qk = 0;
  // integration order on a tensor element whose Jacobian determinant has polynomial order order_w
  static int Get(int order_w) { return 2 * p_trial + (q_order_jac ? order_w : 0) + q_order_extra_qk; }
  // Gauss-Legendre points per direction that integrate that order exactly
  static int GetQ1d(int order_w) { return Get(order_w) / 2 + 1; }
};

}  // namespace fem

// ---- mesh: the geometry factor data of one block of hexahedra --------------------------------------------------------
class Mesh {
  pa_geom *geom_ = nullptr;
  int ne_, q1d_, mesh_order_;
  int nq_dense_ = 0;  // > 0: element block described by dense tables (tetrahedra, ...), that many quadrature points
  int dim_ = 3, sdim_ = 3;
  // host copy of the vertices: corner nodes per element (tensor blocks: lexicographic corners; dense blocks: the first
  // dim + 1 nodes of a simplex) and the node coordinates -- what the AMS set-up reads from the ParMesh (hypre/ams.cpp:64-100)
  int ncorner_ = 0;
  std::vector<int32_t> corner_nodes_;
  std::vector<double> nodes_;
  // this mesh as the uniform refinement of another one (SetRefinementTransforms)
  const Mesh *parent_ = nullptr;
  std::vector<int32_t> embed_parent_, embed_matrix_;
  int n_point_matrices_ = 0;
  std::vector<double> point_matrices_;

public:
  // node_offsets [ne][(mesh_order + 1)^3] lattice order, nodes [num_nodes][3], attr [ne] (1-based); the quadrature is
  // the tensor Gauss-Legendre rule with q1d points (fem::DefaultIntegrationOrder::GetQ1d)
  Mesh(const Context &ctx, int num_elem, int mesh_order, int num_nodes, const int32_t *node_offsets, const double *nodes,
       const int32_t *attr, int q1d);
  // any element type: what AssembleCeedGeometryData gets for a non-tensor block (pa_geom_create_dense)
  Mesh(const Context &ctx, const pa_mesh_dense_desc &desc);
  Mesh(const Mesh &) = delete;
  ~Mesh();
  bool IsDense() const { return nq_dense_ > 0; }
  // element and space dimension: 3 / 3, 2 / 2 (plane problems) or 2 / 3 (a block of boundary elements of a 3-D mesh)
  int Dimension() const { return dim_; }
  int SpaceDimension() const { return sdim_; }
  int GetNumQuadraturePoints() const { return nq_dense_ > 0 ? nq_dense_ : q1d_ * q1d_ * q1d_; }
  pa_geom *GetCeedGeomFactorData() const { return geom_; }
  int GetNE() const { return ne_; }
  int GetQ1d() const { return q1d_; }
  int GetMeshOrder() const { return mesh_order_; }
  // coordinates [GetVSize()][SpaceDimension()] of the dofs of a lowest-order H1 space on this mesh (its dofs are the vertices)
  std::vector<double> VertexCoordinates(const class FiniteElementSpace &h1_p1) const;
  // This mesh is a refinement of `parent` (utils/geodata.cpp:426-460 keeps every uniformly refined mesh of the sequence as a
  // multigrid level): what mfem::Mesh::GetRefinementTransforms() returns -- embeddings[e] = {parent element, matrix} and the
  // point matrices [nmat][vertices per element][Dimension()]: the vertices of the child in the PARENT's reference coordinates
  // (tensor blocks: the eight corners in lexicographic order).  FiniteElementSpaceHierarchy builds the prolongation between
  // two levels on the two meshes from it (fespace.cpp:246-251).
  void SetRefinementTransforms(const Mesh &parent, const int32_t *embed_parent, const int32_t *embed_matrix, int nmat,
                               const double *point_matrices);
  const Mesh *GetParent() const { return parent_; }
  int GetNumPointMatrices() const { return n_point_matrices_; }
  int GetNumCorners() const { return ncorner_; }
  const std::vector<int32_t> &GetEmbeddingParents() const { return embed_parent_; }
  const std::vector<int32_t> &GetEmbeddingMatrices() const { return embed_matrix_; }
  const std::vector<double> &GetPointMatrices() const { return point_matrices_; }
};

// ---- finite element spaces ------------------------------------------------------------------------------------------
class FiniteElementSpace {
  const Context *ctx_;
  const Mesh *mesh_;
  int fe_type_, order_, elem_size_, vsize_, true_vsize_;
  std::vector<int32_t> offsets_, dof_map_;
  std::vector<uint8_t> orients_;
  std::vector<double> Bc_, Gc_, Bo_;
  std::vector<int8_t> curl_orients_;     // dense spaces: the tridiagonal dof transformation (ND tetrahedra, p >= 2)
  std::vector<double> interp_, deriv_;   // dense spaces: [qcomp Q][P] value and [3 Q][P] curl / gradient tables
  std::vector<int32_t> ess_tdofs_;
  const Halo *halo_;
  mutable std::map<const FiniteElementSpace *, std::unique_ptr<Operator>> G_;

public:
  // fe_type PA_FE_HCURL | PA_FE_H1; offsets [ne][P] into the local (L-) vector, orients (HCURL) the sign flips,
  // dof_map the tensor -> native local ordering (NULL: lexicographic).  n_true < 0: one rank, every dof is a true dof
  FiniteElementSpace(const Context &ctx, const Mesh &mesh, int fe_type, int order, int vsize, const int32_t *offsets,
                     const uint8_t *orients, const int32_t *dof_map, int n_true = -1, const Halo *halo = nullptr);
  // The same for a space given by dense tables on a dense Mesh (fem/libceed/basis.cpp:40-85, restriction.cpp:207-385):
  // fe_type PA_FE_HCURL | PA_FE_H1 | PA_FE_HDIV, elem_size dofs per element, interp [qcomp Q][P], deriv [3 Q][P] or NULL,
  // orients or curl_orients [ne][P][3] (not both)
  FiniteElementSpace(const Context &ctx, const Mesh &mesh, int fe_type, int order, int elem_size, int vsize,
                     const int32_t *offsets, const uint8_t *orients, const int8_t *curl_orients, const double *interp,
                     const double *deriv, int n_true = -1, const Halo *halo = nullptr);
  bool IsDense() const { return !interp_.empty() || !deriv_.empty(); }
  pa_dense_basis_desc GetCeedDenseBasis() const;
  const Context &GetContext() const { return *ctx_; }
  const Mesh &GetMesh() const { return *mesh_; }
  int GetFEType() const { return fe_type_; }
  int GetMaxElementOrder() const { return order_; }
  int GetVSize() const { return vsize_; }
  int GetTrueVSize() const { return true_vsize_; }
  int GetElemSize() const { return elem_size_; }
  // local dof of tensor (lexicographic) index t of element e (dense spaces: of native index t)
  int32_t GetElementDof(int e, int t) const {
    const int j = dof_map_.empty() ? t : (dof_map_[t] >= 0 ? dof_map_[t] : -1 - dof_map_[t]);
    return offsets_[(size_t)e * elem_size_ + j];
  }
  const Halo *GetHalo() const { return halo_; }
  pa_restriction_desc GetCeedElemRestriction() const;
  pa_basis_desc GetCeedBasis() const;
  // essential true dofs of the boundary conditions on this space (what FiniteElementSpace::GetEssentialTrueDofs
  // extracts from boundary attribute markers with MFEM)
  void SetEssentialTrueDofs(const int32_t *tdofs, int n) { ess_tdofs_.assign(tdofs, tdofs + n); }
  const std::vector<int32_t> &GetEssentialTrueDofs() const { return ess_tdofs_; }
  // discrete gradient from the H1 space `aux` of the same order into this Nedelec space (fespace.cpp:171-186)
  const Operator &GetDiscreteInterpolator(const FiniteElementSpace &aux) const;
  // dof and sign (true: flipped) of tensor (lexicographic) index t of element e -- native index t for dense spaces
  std::pair<int32_t, bool> GetElementDofSigned(int e, int t) const;
  // Dense spaces on a refined mesh: the local interpolation matrices [GetNumPointMatrices()][P][P] of this space's element for
  // the mesh's point matrices (mfem::FiniteElement::GetLocalInterpolation; row = fine dof, column = parent dof, native order).
  // Tensor spaces compute theirs from the 1-D bases and need none.
  void SetLocalInterpolation(const double *M, int nmat);
  const std::vector<double> &GetLocalInterpolation() const { return local_interp_; }

private:
  std::vector<double> local_interp_;
};

// vdim copies of a scalar space -- mfem::FiniteElementSpace(mesh, fec, vdim, ordering) as far as GradientIntegrator needs it: the
// libCEED restriction of fem/libceed/restriction.cpp:137-142 (offsets of component 0, times vdim for Ordering::byVDIM, and the
// component stride: 1 for byVDIM, the number of dofs for byNODES)
class VectorFiniteElementSpace {
  const FiniteElementSpace *scalar_;
  int vdim_;
  bool by_vdim_;
  std::vector<int32_t> offsets_;

public:
  VectorFiniteElementSpace(const FiniteElementSpace &scalar, int vdim, bool by_vdim = false);
  const FiniteElementSpace &GetScalarSpace() const { return *scalar_; }
  int GetVDim() const { return vdim_; }
  int GetVSize() const { return vdim_ * scalar_->GetVSize(); }
  int GetCompStride() const { return by_vdim_ ? 1 : scalar_->GetVSize(); }
  pa_restriction_desc GetCeedElemRestriction() const;  // one component; lsize = GetVSize()
};

class FiniteElementSpaceHierarchy {
  std::vector<std::unique_ptr<FiniteElementSpace>> fespaces_;
  mutable std::vector<std::unique_ptr<Operator>> P_;
  const Operator &BuildProlongationAtLevel(std::size_t l) const;

public:
  FiniteElementSpaceHierarchy() = default;
  explicit FiniteElementSpaceHierarchy(std::unique_ptr<FiniteElementSpace> &&fespace) { AddLevel(std::move(fespace)); }
  std::size_t GetNumLevels() const { return fespaces_.size(); }
  void AddLevel(std::unique_ptr<FiniteElementSpace> &&fespace) {
    fespaces_.push_back(std::move(fespace));
    P_.push_back(nullptr);
  }
  FiniteElementSpace &GetFESpaceAtLevel(std::size_t l) { return *fespaces_.at(l); }
  const FiniteElementSpace &GetFESpaceAtLevel(std::size_t l) const { return *fespaces_.at(l); }
  FiniteElementSpace &GetFinestFESpace() { return *fespaces_.back(); }
  const FiniteElementSpace &GetFinestFESpace() const { return *fespaces_.back(); }
  const Operator &GetProlongationAtLevel(std::size_t l) const { return P_.at(l) ? *P_[l] : BuildProlongationAtLevel(l); }
  std::vector<const Operator *> GetProlongationOperators() const;
  std::vector<const Operator *> GetDiscreteInterpolators(const FiniteElementSpaceHierarchy &aux_fespaces) const;
};

// ---- integrators ----------------------------------------------------------------------------------------------------
class BilinearFormIntegrator {
protected:
  const MaterialPropertyCoefficient *Q;
  bool transpose;
  // adds one sub-operator {restriction, basis, QFunction, coefficient context, eval modes} to `op`
  static void AssembleCeedOperator(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test, int qf,
                                   const std::vector<double> &ctx, int trial_ops, int test_ops);

public:
  explicit BilinearFormIntegrator(const MaterialPropertyCoefficient *Q = nullptr, bool transpose = false)
      : Q(Q), transpose(transpose) {}
  explicit BilinearFormIntegrator(const MaterialPropertyCoefficient &Q, bool transpose = false) : Q(&Q), transpose(transpose) {}
  virtual ~BilinearFormIntegrator() = default;
  virtual void Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const = 0;
};
#define PA_DECLARE_INTEGRATOR(Name)                                                                       \
  class Name : public BilinearFormIntegrator {                                                            \
  public:                                                                                                 \
    using BilinearFormIntegrator::BilinearFormIntegrator;                                                 \
    void Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const override; \
  }
PA_DECLARE_INTEGRATOR(MassIntegrator);          // H1, (Q u, v)                      fem/integ/mass.cpp
PA_DECLARE_INTEGRATOR(VectorFEMassIntegrator);  // H(curl) / H(div) and the two mixed pairs, (Q u, v)   fem/integ/vecfemass.cpp
PA_DECLARE_INTEGRATOR(DiffusionIntegrator);     // H1, (Q grad u, grad v)            fem/integ/diffusion.cpp
PA_DECLARE_INTEGRATOR(CurlCurlIntegrator);      // H(curl), (Q curl u, curl v)       fem/integ/curlcurl.cpp:23-75
PA_DECLARE_INTEGRATOR(DivDivIntegrator);        // H(div), (Q div u, div v)          fem/integ/divdiv.cpp
PA_DECLARE_INTEGRATOR(MixedVectorGradientIntegrator);  // H1 x H(curl) | H(div), (Q grad u, v)  fem/integ/mixedvecgrad.cpp
PA_DECLARE_INTEGRATOR(MixedVectorWeakDivergenceIntegrator);  // H(curl) x H1, -(Q u, grad v)  fem/integ/mixedvecgrad.cpp:146-202
PA_DECLARE_INTEGRATOR(MixedVectorCurlIntegrator);      // H(curl) x H(curl) | H(div), (Q curl u, v)    fem/integ/mixedveccurl.cpp:21-73
PA_DECLARE_INTEGRATOR(MixedVectorWeakCurlIntegrator);  // H(curl) | H(div) x H(curl), -(Q u, curl v)   fem/integ/mixedveccurl.cpp:75-120
#undef PA_DECLARE_INTEGRATOR
// (H1)^d, (Q u, v) with a d x d coefficient: MassIntegrator with num_comp = 2 | 3 components (fem/integ/mass.cpp:35-48,
// f_apply_h1_2 | _3); like GradientIntegrator below it takes a VectorFiniteElementSpace and builds its operator directly
class VectorMassIntegrator : public BilinearFormIntegrator {
public:
  using BilinearFormIntegrator::BilinearFormIntegrator;
  void Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const override;  // (the scalar form)
  void Assemble(pa_op *op, const VectorFiniteElementSpace &fes) const;
  std::unique_ptr<ceed::Operator> PartialAssemble(const VectorFiniteElementSpace &fes) const;
};
// H1 x (H1)^d, (Q grad u, v): fem/integ/grad.cpp:16-72 (f_apply_hcurlh1d_*).  The test space has space_dim components, which
// BilinearForm's scalar spaces do not describe: the operator is built directly, height = test.GetVSize(), width = trial.GetVSize()
class GradientIntegrator : public BilinearFormIntegrator {
public:
  using BilinearFormIntegrator::BilinearFormIntegrator;
  void Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const override;  // (throws: see below)
  void Assemble(pa_op *op, const FiniteElementSpace &trial, const VectorFiniteElementSpace &test) const;
  std::unique_ptr<ceed::Operator> PartialAssemble(const FiniteElementSpace &trial, const VectorFiniteElementSpace &test) const;
};
#define PA_DECLARE_INTEGRATOR2(Name)                                                                      \
  class Name : public BilinearFormIntegrator {                                                            \
    const MaterialPropertyCoefficient *Q_mass;                                                            \
    bool transpose_mass;                                                                                  \
                                                                                                          \
  public:                                                                                                 \
    Name(const MaterialPropertyCoefficient &Q, const MaterialPropertyCoefficient &Q_mass, bool transpose = false, \
         bool transpose_mass = false)                                                                     \
        : BilinearFormIntegrator(Q, transpose), Q_mass(&Q_mass), transpose_mass(transpose_mass) {}        \
    void Assemble(pa_op *op, const FiniteElementSpace &trial, const FiniteElementSpace &test) const override; \
  }
PA_DECLARE_INTEGRATOR2(DiffusionMassIntegrator);  // H1, (Q grad u, grad v) + (Q_mass u, v)     fem/integ/diffusionmass.cpp
PA_DECLARE_INTEGRATOR2(CurlCurlMassIntegrator);   // H(curl), (Q curl u, curl v) + (Q_mass u, v) fem/integ/curlcurlmass.cpp:16-68
PA_DECLARE_INTEGRATOR2(DivDivMassIntegrator);     // H(div), (Q div u, div v) + (Q_mass u, v)     fem/integ/divdivmass.cpp
#undef PA_DECLARE_INTEGRATOR2

// An assembled local operator that owns its matrix (what FullAssemble returns; hypre::HypreCSRMatrix in the reference)
class CsrMatrix : public CsrOperator {
  pa_csr *owned_;

public:
  CsrMatrix(const Context &ctx, pa_csr *m) : CsrOperator(ctx, m), owned_(m) {}
  ~CsrMatrix() override { pa_csr_destroy(owned_); }
};

class BilinearForm {
protected:
  const FiniteElementSpace &trial_fespace, &test_fespace;
  std::vector<std::unique_ptr<BilinearFormIntegrator>> domain_integs;
  std::vector<std::pair<const FiniteElementSpace *, std::unique_ptr<BilinearFormIntegrator>>> boundary_integs;
  std::unique_ptr<ceed::Operator> PartialAssemble(const FiniteElementSpace &trial, const FiniteElementSpace &test) const;

public:
  inline static int pa_order_threshold = 1;  // order below which Assemble returns a matrix (bilinearform.hpp:36)
  BilinearForm(const FiniteElementSpace &trial_fespace, const FiniteElementSpace &test_fespace)
      : trial_fespace(trial_fespace), test_fespace(test_fespace) {}
  explicit BilinearForm(const FiniteElementSpace &fespace) : BilinearForm(fespace, fespace) {}
  const FiniteElementSpace &GetTrialSpace() const { return trial_fespace; }
  const FiniteElementSpace &GetTestSpace() const { return test_fespace; }
  template <typename T, typename... U>
  void AddDomainIntegrator(U &&...args) {
    domain_integs.push_back(std::make_unique<T>(std::forward<U>(args)...));
  }
  // Boundary integrators act on a block of boundary elements: `bdr_fespace` is the view of the form's space on that block
  // (a dense-table space on a Mesh with Dimension() = 2, SpaceDimension() = 3 whose restriction indexes the same L-vector --
  // what FiniteElementSpace::GetCeedElemRestriction(ceed, geom, indices) returns for boundary elements,
  // fem/libceed/restriction.cpp:15-111); bilinearform.cpp:60-100 adds them as further sub-operators of the same operator.
  template <typename T, typename... U>
  void AddBoundaryIntegrator(const FiniteElementSpace &bdr_fespace, U &&...args) {
    boundary_integs.emplace_back(&bdr_fespace, std::make_unique<T>(std::forward<U>(args)...));
  }
  std::unique_ptr<ceed::Operator> PartialAssemble() const { return PartialAssemble(trial_fespace, test_fespace); }
  std::unique_ptr<CsrMatrix> FullAssemble(bool skip_zeros) const { return FullAssemble(*PartialAssemble(), skip_zeros); }
  static std::unique_ptr<CsrMatrix> FullAssemble(const ceed::Operator &op, bool skip_zeros);
  std::unique_ptr<Operator> Assemble(bool skip_zeros) const;
  // one operator per level l0..L-1: the coarsest requested level is assembled, the others reuse its quadrature data
  // through CeedOperatorCoarsen (bilinearform.cpp:153-201)
  std::vector<std::unique_ptr<Operator>> Assemble(const FiniteElementSpaceHierarchy &fespaces, bool skip_zeros,
                                                  std::size_t l0 = 0) const;
};

namespace ceed {
// The operator of `op_fine` on another space of the same mesh, reusing its quadrature data (operator.cpp:525-585)
std::unique_ptr<Operator> CeedOperatorCoarsen(const Operator &op_fine, const FiniteElementSpace &fespace_coarse);
}  // namespace ceed

// ParOperator that owns its local operator and takes sizes / halo from the space (rap.hpp:24-110); the essential dofs
// are set afterwards like in the reference (SetEssentialTrueDofs, rap.cpp:38-54).
class FespaceParOperator : public Operator {
  const Context *ctx_;
  std::unique_ptr<Operator> local_;
  const FiniteElementSpace *fespace_;
  std::unique_ptr<ParOperator> par_;
  std::vector<int32_t> ess_tdofs_;

public:
  const std::vector<int32_t> &GetEssentialTrueDofsHost() const { return ess_tdofs_; }
  FespaceParOperator(std::unique_ptr<Operator> &&A, const FiniteElementSpace &fespace);
  void SetEssentialTrueDofs(const std::vector<int32_t> &tdofs, ParOperator::DiagonalPolicy policy);
  const ParOperator &Par() const { return *par_; }
  const Operator &LocalOperator() const { return *local_; }
  const FiniteElementSpace &GetFESpace() const { return *fespace_; }
  void Mult(const Vector &x, Vector &y) const override { par_->Mult(x, y); }
  void MultTranspose(const Vector &x, Vector &y) const override { par_->MultTranspose(x, y); }
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override { par_->AddMult(x, y, a); }
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override { par_->AddMultTranspose(x, y, a); }
  void AssembleDiagonal(Vector &diag) const override { par_->AssembleDiagonal(diag); }
  bool IsSymmetric() const override { return par_->IsSymmetric(); }
};

// The operators of all multigrid levels (and of the auxiliary space), finest last (linalg/operator.hpp:424-493)
class MultigridOperator : public Operator {
  std::vector<std::unique_ptr<FespaceParOperator>> ops_, aux_ops_;

public:
  explicit MultigridOperator(std::size_t l) {
    ops_.reserve(l);
    aux_ops_.reserve(l);
  }
  void AddOperator(std::unique_ptr<FespaceParOperator> &&op) {
    ops_.push_back(std::move(op));
    height = ops_.back()->Height(), width = ops_.back()->Width();
  }
  void AddAuxiliaryOperator(std::unique_ptr<FespaceParOperator> &&aux_op) { aux_ops_.push_back(std::move(aux_op)); }
  bool HasAuxiliaryOperators() const { return !aux_ops_.empty(); }
  std::size_t GetNumLevels() const { return ops_.size(); }
  std::size_t GetNumAuxiliaryLevels() const { return aux_ops_.size(); }
  const FespaceParOperator &GetFinestOperator() const { return *ops_.back(); }
  const FespaceParOperator &GetFinestAuxiliaryOperator() const { return *aux_ops_.back(); }
  const FespaceParOperator &GetOperatorAtLevel(std::size_t l) const { return *ops_.at(l); }
  const FespaceParOperator &GetAuxiliaryOperatorAtLevel(std::size_t l) const { return *aux_ops_.at(l); }
  void Mult(const Vector &x, Vector &y) const override { GetFinestOperator().Mult(x, y); }
  void MultTranspose(const Vector &x, Vector &y) const override { GetFinestOperator().MultTranspose(x, y); }
  void AddMult(const Vector &x, Vector &y, double a = 1.0) const override { GetFinestOperator().AddMult(x, y, a); }
  void AddMultTranspose(const Vector &x, Vector &y, double a = 1.0) const override {
    GetFinestOperator().AddMultTranspose(x, y, a);
  }
  void AssembleDiagonal(Vector &diag) const override { GetFinestOperator().AssembleDiagonal(diag); }
  bool IsSymmetric() const override { return GetFinestOperator().IsSymmetric(); }
};

}  // namespace palace
