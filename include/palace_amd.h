/*
 * palace_amd.h — C ABI of the MI355X-native partial-assembly operator library (libpalace_amd.so).
 *
 * This is the drop-in boundary for Palace's libCEED glue: every entry point replaces one libCEED
 * call sequence that `palace/fem/libceed/` issues today.  Citations are file:line in the
 * reference tree (awslabs/palace).  Rules of the ABI:
 *   - plain C: pointers, sizes, enums; no C++/torch/mfem types; no exceptions cross it;
 *   - every function returns 0 on success, non-zero on error; the message is retrieved with
 *     pa_last_error() (reference convention: int codes + CeedGetErrorMessage, libceed/ceed.hpp:13-33);
 *   - descriptor arrays are HOST pointers and are copied at creation (the reference passes
 *     CEED_COPY_VALUES: restriction.cpp:195,367, integrator.cpp:447-448);
 *   - x / y vectors are DEVICE pointers owned by the caller, never copied (CEED_USE_POINTER,
 *     operator.cpp:170-176); work is enqueued on the hipStream_t passed as `void *stream` and is
 *     asynchronous with respect to the host;
 *   - one apply at a time per operator (the reference's Mult is not re-entrant either,
 *     operator.hpp:38).
 */
#ifndef PALACE_AMD_H
#define PALACE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_geom pa_geom; /* ceed::CeedGeomFactorData, fem/mesh.hpp:27-69 (shared, ref-counted) */
typedef struct pa_op pa_op;     /* palace::ceed::Operator,   fem/libceed/operator.hpp:32-65           */

/* Evaluation-mode bitmask, identical to palace::ceed::EvalMode (fem/libceed/integrator.hpp:15-23). */
enum pa_eval_mode {
  PA_EVAL_WEIGHT = 1 << 0,
  PA_EVAL_NONE = 1 << 1,
  PA_EVAL_INTERP = 1 << 2,
  PA_EVAL_GRAD = 1 << 3,
  PA_EVAL_DIV = 1 << 4,
  PA_EVAL_CURL = 1 << 5
};

/* QFunctions (the pointwise D stage).  Each id names the reference QFunction it reproduces. */
enum pa_qfunction {
  PA_QF_HDIV_33 = 0,      /* f_apply_hdiv_33      fem/qfunctions/33/hdiv_33_qf.h:10-30      curl-curl        */
  PA_QF_HCURL_33 = 1,     /* f_apply_hcurl_33     fem/qfunctions/33/hcurl_33_qf.h:10-28     ND mass, H1 diffusion */
  PA_QF_HDIVMASS_33 = 2,  /* f_apply_hdivmass_33  fem/qfunctions/33/hdivmass_33_qf.h:10-44  curl-curl + mass */
  PA_QF_HCURLMASS_33 = 3, /* f_apply_hcurlmass_33 fem/qfunctions/33/hcurlmass_33_qf.h       H1 diffusion + mass */
  PA_QF_H1_1 = 4,         /* f_apply_h1_1         fem/qfunctions/1/h1_1_qf.h                H1 mass          */
  /* 2-D (space_dim = dim = 2; dense-table path only), fem/integ/curlcurl.cpp:40-47,65-68 */
  PA_QF_HCURL_22 = 5,     /* f_apply_hcurl_22     fem/qfunctions/22/hcurl_22_qf.h:10-30     ND mass          */
  PA_QF_L2_1 = 6,         /* f_apply_l2_1         fem/qfunctions/1/l2_1_qf.h:10-24          curl-curl (scalar curl, q_w input) */
  PA_QF_HDIVMASS_22 = 7,  /* f_apply_hdivmass_22  fem/qfunctions/22/hdivmass_22_qf.h:11-37  curl-curl + mass */
  /* boundary elements (dim = 2 in space_dim = 3; dense-table path only) */
  PA_QF_HCURL_32 = 8,     /* f_apply_hcurl_32     fem/qfunctions/32/hcurl_32_qf.h:10-30     ND surface mass (impedance,
                             absorbing and lumped-port boundary terms) */
  /* mixed H(curl) / H(div) Piola maps on one H(curl) space (the Floquet-periodic terms of SpaceOperator,
   * models/spaceoperator.cpp:305-309); tensor-product hexahedra, matrix-free D */
  PA_QF_HCURLHDIV_33 = 9, /* f_apply_hcurlhdiv_33 fem/qfunctions/33/hcurlhdiv_33_qf.h:10-31 MixedVectorWeakCurlIntegrator:
                             (C u, curl v), trial Interp, test Curl */
  PA_QF_HDIVHCURL_33 = 10, /* f_apply_hdivhcurl_33 fem/qfunctions/33/hcurlhdiv_33_qf.h:33-54 MixedVectorCurlIntegrator:
                             (C curl u, v), trial Curl, test Interp.  Both also with two different spaces, Interp / Interp:
                             pa_op_add_sub_dense_mixed */
  /* element error integrators (pa_error_op_create), two inputs with the Piola maps of their spaces */
  PA_QF_HCURLHDIV_ERROR_33 = 11, /* f_apply_hcurlhdiv_error_33 fem/qfunctions/33/hcurlhdiv_error_33_qf.h:10-43 */
  PA_QF_HDIVHCURL_ERROR_33 = 12, /* f_apply_hdivhcurl_error_33 fem/qfunctions/33/hcurlhdiv_error_33_qf.h:45-78 */
  /* the remaining 2-D and boundary-element (2-D elements in 3-D space) forms, dense tables; the boundary ones on geometry
   * data with space_dim = 3.  On boundary elements PA_QF_L2_1 (surface curl-curl) and PA_QF_H1_1 (H1 mass) are the same
   * QFunctions as in the plane: they only read w detJ */
  PA_QF_HDIVMASS_32 = 13,  /* f_apply_hdivmass_32  fem/qfunctions/32/hdivmass_32_qf.h   ND boundary curl-curl + mass */
  PA_QF_HCURLMASS_22 = 14, /* f_apply_hcurlmass_22 fem/qfunctions/22/hcurlmass_22_qf.h  2-D H1 diffusion + mass */
  PA_QF_HCURLMASS_32 = 15, /* f_apply_hcurlmass_32 fem/qfunctions/32/hcurlmass_32_qf.h  H1 boundary diffusion + mass;
                              H1 diffusion alone: PA_QF_HCURL_22 / PA_QF_HCURL_32 with EVAL_GRAD (fem/integ/diffusion.cpp) */
  /* line elements (pa_mesh_dense_desc dim = 1, space_dim = 2 or 3: boundaries of plane problems, curves in space), dense tables:
   * ND mass with Interp, H1 diffusion with Grad (one derivative), H1 diffusion + mass; H1 mass is PA_QF_H1_1 */
  PA_QF_HCURL_21 = 16,     /* f_apply_hcurl_21     fem/qfunctions/21/hcurl_21_qf.h:10-29 */
  PA_QF_HCURL_31 = 17,     /* f_apply_hcurl_31     fem/qfunctions/31/hcurl_31_qf.h */
  PA_QF_HCURLMASS_21 = 18, /* f_apply_hcurlmass_21 fem/qfunctions/21/hcurlmass_21_qf.h */
  PA_QF_HCURLMASS_31 = 19, /* f_apply_hcurlmass_31 fem/qfunctions/31/hcurlmass_31_qf.h:10-38 */
  /* two spaces on plane elements (pa_op_add_sub_dense_mixed, pa_error_op_create with 2-D geometry data) */
  PA_QF_HCURLHDIV_22 = 20,       /* f_apply_hcurlhdiv_22       fem/qfunctions/22/hcurlhdiv_22_qf.h:10-30 */
  PA_QF_HDIVHCURL_22 = 21,       /* f_apply_hdivhcurl_22       fem/qfunctions/22/hcurlhdiv_22_qf.h:32-52 */
  PA_QF_HCURLHDIV_ERROR_22 = 22, /* f_apply_hcurlhdiv_error_22 fem/qfunctions/22/hcurlhdiv_error_22_qf.h:10-41 */
  PA_QF_HDIVHCURL_ERROR_22 = 23, /* f_apply_hdivhcurl_error_22 fem/qfunctions/22/hcurlhdiv_error_22_qf.h:43-74 */
  PA_QF_HDIV_22 = 24,            /* f_apply_hdiv_22 fem/qfunctions/22/hdiv_22_qf.h:10-30: mass of a plane H(div) space
                                    (pa_op_add_sub_dense with PA_FE_HDIV, Interp) */
  PA_QF_L2H1_ERROR = 25,         /* f_apply_l2h1_error fem/qfunctions/l2h1_error_qf.h:14-30: element error between two scalar
                                    fields (the scalar curl of a plane field and its H1 recovery, errorestimator.cpp:466-472);
                                    pa_error_op_create with two scalar (PA_FE_H1 descriptor) bases, 1 x 1 pair context */
  /* The members of the 32 | 31 | 21 families with the contravariant map, the div-div + mass pair and the gradient form.  No
   * integrator that Palace's drivers add uses them (fem/integ/{vecfemass,mixedvecgrad,divdivmass,grad}.cpp select them for
   * H(div) spaces on boundary / line elements, DivDivMassIntegrator and GradientIntegrator); built for completeness of the
   * QFunction table, on the dense path (PA_FE_HDIV descriptors; pa_op_add_sub_dense / _mixed / _gradient). */
  PA_QF_HDIV_32 = 26,      /* f_apply_hdiv_32      fem/qfunctions/32/hdiv_32_qf.h:10-31  H(div) mass, boundary elements */
  PA_QF_HDIV_21 = 27,      /* f_apply_hdiv_21      fem/qfunctions/21/hdiv_21_qf.h        ... line elements in the plane */
  PA_QF_HDIV_31 = 28,      /* f_apply_hdiv_31      fem/qfunctions/31/hdiv_31_qf.h:10-28  ... line elements in space */
  PA_QF_L2MASS_22 = 29,    /* f_apply_l2mass_22    fem/qfunctions/22/l2mass_22_qf.h      div-div + H(div) mass (Interp | Div |
                              Weight; pair context: space_dim x space_dim mass coefficient first, then the scalar one) */
  PA_QF_L2MASS_33 = 30,    /* f_apply_l2mass_33    fem/qfunctions/33/l2mass_33_qf.h:10-42 */
  PA_QF_L2MASS_32 = 31,    /* f_apply_l2mass_32    fem/qfunctions/32/l2mass_32_qf.h:10-42 */
  PA_QF_L2MASS_21 = 32,    /* f_apply_l2mass_21    fem/qfunctions/21/l2mass_21_qf.h:10-40 */
  PA_QF_L2MASS_31 = 33,    /* f_apply_l2mass_31    fem/qfunctions/31/l2mass_31_qf.h */
  PA_QF_HCURLHDIV_32 = 34, /* f_apply_hcurlhdiv_32 fem/qfunctions/32/hcurlhdiv_32_qf.h:10-30  two spaces (pa_op_add_sub_dense_mixed) */
  PA_QF_HDIVHCURL_32 = 35, /* f_apply_hdivhcurl_32 fem/qfunctions/32/hcurlhdiv_32_qf.h:32-52 */
  PA_QF_HCURLHDIV_21 = 36, /* f_apply_hcurlhdiv_21 fem/qfunctions/21/hcurlhdiv_21_qf.h */
  PA_QF_HDIVHCURL_21 = 37,
  PA_QF_HCURLHDIV_31 = 38, /* f_apply_hcurlhdiv_31 fem/qfunctions/31/hcurlhdiv_31_qf.h:10-28 */
  PA_QF_HDIVHCURL_31 = 39, /* f_apply_hdivhcurl_31 fem/qfunctions/31/hcurlhdiv_31_qf.h:30-48 */
  PA_QF_HCURLH1D_22 = 40,  /* f_apply_hcurlh1d_22  fem/qfunctions/22/hcurlh1d_22_qf.h  (C grad u, v), v in a vector H1 space */
  PA_QF_HCURLH1D_33 = 41,  /* f_apply_hcurlh1d_33  fem/qfunctions/33/hcurlh1d_33_qf.h:10-30 (pa_op_add_sub_dense_gradient) */
  PA_QF_HCURLH1D_32 = 42,  /* f_apply_hcurlh1d_32  fem/qfunctions/32/hcurlh1d_32_qf.h:10-31 */
  PA_QF_HCURLH1D_21 = 43,  /* f_apply_hcurlh1d_21  fem/qfunctions/21/hcurlh1d_21_qf.h:10-29 */
  PA_QF_HCURLH1D_31 = 44   /* f_apply_hcurlh1d_31  fem/qfunctions/31/hcurlh1d_31_qf.h */
};

enum pa_fe_type {
  PA_FE_H1 = 0,
  PA_FE_HCURL = 1,
  PA_FE_HDIV = 2 /* dense path only: RT mass (Interp + PA_QF_HDIV_33 | _22 | _32 | _21 | _31 by the geometry data, interp = values
                    [dim Q][P]), div-div (Div | Weight + PA_QF_L2_1, fem/integ/divdiv.cpp; deriv = divergence [Q][P]), both in
                    one pass (Interp | Div | Weight + PA_QF_L2MASS_*, fem/integ/divdivmass.cpp); mixed mass with an H(curl) space */
};

/*
 * Element restriction E — what Palace hands to CeedElemRestrictionCreate / ...CreateOriented
 * (fem/libceed/restriction.cpp:364-383 native, :192-203 lexicographic).
 *   offsets[j + elem_size*e] : L-vector index of local dof j of element e (native local order)
 *   orients[j + elem_size*e] : non-zero => u_e[j] = -x[offset]  (may be NULL: no flips)
 */
typedef struct {
  int32_t num_elem;
  int32_t elem_size;
  int32_t lsize;
  const int32_t *offsets;
  const uint8_t *orients;
  /* CeedElemRestrictionCreateCurlOriented (restriction.cpp:299-369): the tridiagonal dof
   * transformation of element e, curl_orients[3*(j + elem_size*e) + {0,1,2}] = {sub, main, super}
   * diagonal of row j, dof signs folded in:
   *   u_e[j] = T[3j] x[off[j-1]] + T[3j+1] x[off[j]] + T[3j+2] x[off[j+1]],  E^T applies T^T.
   * NULL unless the space has a non-identity DofTransformation (ND on tetrahedra / prisms, p >= 2).
   * Only the dense-table path (pa_op_add_sub_dense) accepts it; `orients` must then be NULL. */
  const int8_t *curl_orients;
} pa_restriction_desc;

/*
 * Basis B/G on tensor-product hexahedra.  Palace gives libCEED the dense DofToQuad::FULL tables
 * for vector elements (fem/libceed/basis.cpp:40-85) and 1-D tables for scalar tensor elements
 * (:15-38).  The same MFEM element also provides its 1-D closed/open tables
 * (VectorTensorFiniteElement::GetDofToQuad / GetDofToQuadOpen) and its lexicographic->native map
 * (TensorBasisElement::GetDofMap()); this library takes those and uses sum factorisation, so the
 * apply stays HBM-bound instead of dense-GEMM-bound.  The dense tables are optional: when given
 * they are checked at creation against the Kronecker product of the 1-D tables (max-norm 1e-12).
 *
 *   order      p.  HCURL: P = 3 p (p+1)^2 ; H1: P = (p+1)^3
 *   q1d        quadrature points per direction, Q = q1d^3, point index q = qx + q1d (qy + q1d qz)
 *   Bc, Gc     closed (Gauss-Lobatto) basis values / derivatives at the 1-D points, [q1d][p+1]
 *   Bo         open (Gauss-Legendre) basis values, [q1d][p]           (HCURL only)
 *   dof_map    [P] tensor index -> native local index, negative value -1-n = native n with flipped
 *              sign; NULL = identity (H1 lexicographic restriction, restriction.cpp:134-136)
 *   Tensor index of HCURL dofs: x-block i + p (j + (p+1) k), then y-block i + (p+1)(j + p k), then
 *   z-block i + (p+1)(j + (p+1) k), the open direction having p entries.
 *   interp     optional dense [qcomp*Q][P] (native dof order), deriv optional dense curl/grad [3*Q][P]
 */
typedef struct {
  int32_t fe_type;
  int32_t order;
  int32_t q1d;
  const double *Bc;
  const double *Gc;
  const double *Bo;
  const int32_t *dof_map;
  const double *interp;
  const double *deriv;
} pa_basis_desc;

/*
 * Mesh geometry — what fem/mesh.cpp:146-209 hands to AssembleCeedGeometryData
 * (fem/libceed/integrator.cpp:335-421): nodal coordinates of a tensor H1 mesh space of order
 * `mesh_order` (2 for the 27-node hexes), its element restriction, its 1-D tables at the
 * quadrature points, the quadrature weights and the per-element (1-based, local) attribute.
 *   node_offsets[n + npe*e]  node id of lattice node n = i + (mesh_order+1) (j + (mesh_order+1) k)
 *   nodes[3*id + c]          coordinate c of node id (byVDIM ordering)
 *   mesh_B, mesh_G           [q1d][mesh_order+1] nodal basis values / derivatives at the 1-D points
 *   qweight1d                [q1d]; the point weight is the product of the three 1-D weights
 */
typedef struct {
  int32_t num_elem;
  int32_t mesh_order;
  int32_t q1d;
  int32_t num_nodes;
  const int32_t *node_offsets;
  const double *nodes;
  const int32_t *attr;
  const double *mesh_B;
  const double *mesh_G;
  const double *qweight1d;
} pa_mesh_desc;

/*
 * Basis B/G of a non-tensor element (tetrahedra, prisms, ... and hexahedra when the caller only has
 * the dense tables): exactly what InitNonTensorBasis hands to CeedBasisCreateHcurl / CeedBasisCreateH1
 * (fem/libceed/basis.cpp:40-85), from MFEM's DofToQuad::FULL tables.
 *   interp [qcomp*Q][P] row-major, row = c*Q + q ; qcomp = dim for HCURL, 1 for H1
 *   deriv  [3*Q][P]     curl (HCURL) or gradient (H1) in reference coordinates; [Q][P] for the scalar
 *                       curl of 2-D HCURL elements
 */
typedef struct {
  int32_t fe_type;
  int32_t num_dofs;  /* P */
  int32_t num_qpts;  /* Q */
  const double *interp;
  const double *deriv;
} pa_dense_basis_desc;

/*
 * Mesh geometry of a non-tensor element block: nodal H1 mesh space given by the dense gradient
 * table of its nodal basis at the quadrature points (the non-tensor branch of what fem/mesh.cpp:146-209
 * passes to AssembleCeedGeometryData).
 *   node_offsets[n + npe*e]   node id of local mesh node n of element e
 *   nodes[3*id + c]           coordinates (byVDIM)
 *   mesh_grad[(d*Q + q)*npe + n]  d phi_n / d xi_d at point q
 *   qweight[q]
 */
typedef struct {
  int32_t num_elem;
  int32_t nodes_per_elem;
  int32_t num_qpts;
  int32_t num_nodes;
  const int32_t *node_offsets;
  const double *nodes;
  const int32_t *attr;
  const double *mesh_grad;
  const double *qweight;
  /* space dimension = element dimension: 3 (default when 0) or 2.  In 2-D nodes are [num_nodes][2],
   * mesh_grad is [2][Q][npe] and the geometry data has 6 rows {attr, w detJ, adj(J)^T/detJ}
   * (fem/qfunctions/22/geom_22_qf.h:9-30). */
  int32_t dim;
  /* space dimension when it differs from the element dimension: 3 with dim = 2 for boundary elements
   * (nodes [num_nodes][3], 8 geometry rows {attr, w detJ, adj(J)^T/detJ (3x2)}, geom_32_qf.h:9-33); 2 or 3 with dim = 1 for
   * line elements (mesh_grad [1][Q][npe], 2 + space_dim rows {attr, w |J|, J / |J|^2}, geom_21_qf.h, geom_31_qf.h);
   * 0 = same as dim. */
  int32_t space_dim;
} pa_mesh_dense_desc;

/* --- library ------------------------------------------------------------------------------- */
const char *pa_last_error(void);
const char *pa_version(void);
/* Number of visible HIP devices (0 => none; every compute entry point then fails loudly). */
int pa_device_count(void);

/* --- geometry factors: replaces AssembleCeedGeometryData + f_build_geom_factor_33
 *     (fem/libceed/integrator.cpp:335-421, fem/qfunctions/33/geom_33_qf.h:9-33).
 *     Result: double[num_elem][11][Q] = {attr, w detJ, adj(J)^T/detJ (col-major)} in HBM. */
int pa_geom_create(const pa_mesh_desc *mesh, void *stream, pa_geom **geom);
/* The same for a non-tensor element block.  The data is kept element-blocked for the MFMA kernel:
 * double[ceil(ne/16)][11][Qpad][16] (Qpad = Q rounded up to 16), entry (e, c, q) at
 * ((e/16 * 11 + c) * Qpad + q) * 16 + e%16; pa_geom_layout reports {ne, Q, Qpad, block}. */
int pa_geom_create_dense(const pa_mesh_dense_desc *mesh, void *stream, pa_geom **geom);
int pa_geom_layout(const pa_geom *geom, int32_t out[4]);
/* Rows of geometry data per point: 11 (3-D), 6 (2-D), 8 (2-D elements in 3-D space). */
int pa_geom_num_rows(const pa_geom *geom);
int pa_geom_retain(pa_geom *geom);
void pa_geom_destroy(pa_geom *geom);
/* Device pointer to the geometry data and its length in doubles (tests / diagnostics).  Tensor blocks keep their elements
 * in an internal (space-filling-curve) order: row p of the data is the caller's element order[p]. */
int pa_geom_element_order(const pa_geom *geom, int32_t *order);
int pa_geom_data(const pa_geom *geom, const double **dev_ptr, size_t *count);

/* --- operator: replaces ceed::Operator (fem/libceed/operator.cpp) ---------------------------- */
/* Operator::Operator(h, w), operator.cpp:17-42. */
int pa_op_create(int32_t height, int32_t width, pa_op **op);
/* AddSubOperator(AssembleCeedOperator(info, ctx, ...)) — operator.cpp:60-87 +
 * fem/libceed/integrator.cpp:423-513.  `ctx` is the CeedIntScalar blob Palace packs in
 * fem/libceed/coefficient.cpp:51-131 (8-byte slots; pair contexts: mass first).  trial_ops /
 * test_ops are pa_eval_mode masks and must be what the reference integrator sets for this
 * QFunction (e.g. Curl|Interp for PA_QF_HDIVMASS_33, fem/integ/curlcurlmass.cpp:55-56). */
int pa_op_add_sub(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                  const pa_basis_desc *basis, int32_t qfunction, const void *ctx, size_t ctx_size,
                  uint32_t trial_ops, uint32_t test_ops);
/* The same with dense basis tables (any element type): E (plain, oriented or curl-oriented), the
 * dense [qcomp*Q x P] contractions on the FP64 matrix cores (v_mfma_f64_16x16x4), D, and the
 * transposes.  `geom` must come from pa_geom_create_dense. */
int pa_op_add_sub_dense(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                        const pa_dense_basis_desc *basis, int32_t qfunction, const void *ctx,
                        size_t ctx_size, uint32_t trial_ops, uint32_t test_ops);
/* Trial space != test space on the same elements: BilinearForm(trial_fespace, test_fespace) with VectorFEMassIntegrator, which
 * picks f_apply_hcurlhdiv_33 (H(curl) trial, H(div) test) or f_apply_hdivhcurl_33 (the other way round) from the map types of
 * the two elements (fem/integ/vecfemass.cpp:88-101) -- the `Flux` operator of FluxProjector (linalg/errorestimator.cpp:164-176).
 * Both evaluation modes are Interp; the H(div) basis carries its value table in `interp`.  An H1 basis on a covariant side
 * enters with its gradient table (`deriv`, Grad): with PA_QF_HCURL_33 / PA_QF_HCURL_22 that is MixedVectorGradientIntegrator
 * (C grad phi, v), H1 trial and H(curl) test (fem/integ/mixedvecgrad.cpp:43-76; models/modeeigensolver.cpp:52), and with
 * PA_QF_HCURLHDIV_* its H(div)-test form (mixedvecgrad.cpp:50-55).  Plane elements: the _22 QFunctions with 2-D geometry
 * data.  PA_QF_HDIV_33 between two spaces: (C curl u, v) with u in H(curl) (its curl table, `deriv`) and v in H(div) --
 * MixedVectorCurlIntegrator with an H(div) test space (fem/integ/mixedveccurl.cpp:41-46); transposed: the weak curl with an H(div)
 * trial space (:88-93; the factor -1 of :111 is the caller's coefficient).  Boundary and line elements: the _32 | _31 | _21 members (PA_QF_HCURL_32 ..., PA_QF_HCURLHDIV_32 ..., PA_QF_HDIVHCURL_32
 * ...; mixedvecgrad.cpp:78-130, vecfemass.cpp).  PA_QF_H1_1 with two scalar bases (PA_FE_H1 descriptors, value tables): MassIntegrator between two scalar spaces, the
 * `Flux` operator of the scalar-flux FluxProjector (errorestimator.cpp:122-160).  op: height = test lsize, width = trial lsize.
 * pa_op_mult_transpose applies the transposed form (test -> trial: the other member of the QFunction pair with the transposed
 * coefficient; Btn = -Atn^T of models/modeeigensolver.cpp:410-418 without assembling).  No essential-dof, diagonal or assembled form. */
int pa_op_add_sub_dense_mixed(pa_op *op, pa_geom *geom, const pa_restriction_desc *trial_restr,
                              const pa_dense_basis_desc *trial_basis, const pa_restriction_desc *test_restr,
                              const pa_dense_basis_desc *test_basis, int32_t qfunction, const void *ctx, size_t ctx_size);
/* GradientIntegrator (fem/integ/grad.cpp:16-72; f_apply_hcurlh1d_22 | _33 | _21 | _31 | _32 by the geometry data): (C grad u, v)
 * with u in a scalar H1 space (trial, Grad: `trial_basis->deriv`) and v in a vector H1 space with space_dim components (test,
 * Interp).  `test_restr` / `test_basis` describe ONE component of the test space -- the scalar value table, offsets = L-vector
 * index of component 0 (already multiplied by the vector dimension for byVDIM ordering, restriction.cpp:137-142) --, component c
 * of a dof lives `comp_stride` entries further (the libCEED comp_stride: 1 for byVDIM, the number of dofs for byNODES), and
 * test_restr->lsize is the size of the whole vector L-vector (= the operator's height).  pa_op_full_assemble gives the rectangular
 * matrix (rows = the dofs of the vector space); no transposed apply, no diagonal. */
int pa_op_add_sub_dense_gradient(pa_op *op, pa_geom *geom, const pa_restriction_desc *trial_restr,
                                 const pa_dense_basis_desc *trial_basis, const pa_restriction_desc *test_restr,
                                 const pa_dense_basis_desc *test_basis, int32_t comp_stride, int32_t qfunction, const void *ctx,
                                 size_t ctx_size);
/* MassIntegrator on a vector-valued H1 space (fem/integ/mass.cpp:35-48: f_apply_h1_2 | _3, fem/qfunctions/{2,3}/h1_*_qf.h;
 * num_comp x num_comp coefficient): `restr` / `basis` describe ONE component as in pa_op_add_sub_dense_gradient (offsets of component
 * 0, comp_stride to the next, restr->lsize = the whole vector = the operator's height and width).  Built from the scalar mass
 * between two scalar spaces (PA_QF_H1_1), one term per non-zero entry of the coefficient: apply, transpose and full assembly; no
 * diagonal.  Volume / plane elements.  (No driver of the reference uses more than one component.) */
int pa_op_add_sub_dense_vector_mass(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr, const pa_dense_basis_desc *basis,
                                    int32_t num_comp, int32_t comp_stride, const void *ctx, size_t ctx_size);
/* AssembleCeedElementErrorIntegrator (fem/libceed/integrator.cpp:550-626) as the flux error estimators use it
 * (linalg/errorestimator.cpp:345-349, :485-489): estimates[e] += int_e |C_2 u_2 - C_1 u_1|^2 for L-vectors u_1, u_2 of two spaces
 * on the elements of `geom`; `ctx` is the pair context PopulateCoefficientContext(dim, first, dim, second) packs.  One value per
 * element, in the element order of `geom`. */
typedef struct pa_error_op pa_error_op;
int pa_error_op_create(pa_geom *geom, const pa_restriction_desc *restr1, const pa_dense_basis_desc *basis1,
                       const pa_restriction_desc *restr2, const pa_dense_basis_desc *basis2, int32_t qfunction,
                       const void *ctx, size_t ctx_size, pa_error_op **out);
int pa_error_op_apply_add(pa_error_op *e, const double *u1, const double *u2, double *estimates, void *stream);
int pa_error_op_num_elem(const pa_error_op *e);
void pa_error_op_destroy(pa_error_op *e);
/* SURVEY.md 8(f)-1, behind BuildParSumOperator (linalg/rap.cpp:843-919) and SpaceOperator::GetSystemMatrix
 * (models/spaceoperator.cpp:786-804): sum_k coeffs[k] * (integrator k) over H(curl) integrators that share the
 * geometry data and the space -- K (PA_QF_HDIV_33), M and C (PA_QF_HCURL_33), K + M (PA_QF_HDIVMASS_33) -- added as ONE
 * sub-operator.  D is linear in the material coefficient, so the sum is a single curl-curl + mass pass whose two
 * contexts are the weighted sums of the terms' contexts per mesh attribute: one pass over the geometry data instead of
 * one per term (a0 K + a2 M for the real part, a1 C for the imaginary part of A(omega) = K + i omega C - omega^2 M).
 * qfunctions[k] / ctxs[k] / ctx_sizes[k] are what pa_op_add_sub would take for term k. */
int pa_op_add_sub_sum(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr, const pa_basis_desc *basis,
                      int32_t nterms, const int32_t *qfunctions, const void *const *ctxs, const size_t *ctx_sizes,
                      const double *coeffs);
int pa_op_add_sub_dense_sum(pa_op *op, pa_geom *geom, const pa_restriction_desc *restr,
                            const pa_dense_basis_desc *basis, int32_t nterms, const int32_t *qfunctions,
                            const void *const *ctxs, const size_t *ctx_sizes, const double *coeffs);
/* Operator::Finalize(), operator.cpp:89-101. */
int pa_op_finalize(pa_op *op);
/* CeedOperatorCoarsen (operator.cpp:525-585): same QFunctions, contexts and geometry data as
 * `fine`, coarse restriction + basis.  `fine` must have a single element block. */
int pa_op_coarsen(const pa_op *fine, const pa_restriction_desc *restr, const pa_basis_desc *basis,
                  pa_op **coarse);
/* The same for an operator made of dense sub-operators (same geometry data, quadrature rule, QFunction
 * and contexts; coarse restriction and dense tables evaluated at the FINE level's quadrature points). */
int pa_op_coarsen_dense(const pa_op *fine, const pa_restriction_desc *restr, const pa_dense_basis_desc *basis,
                        pa_op **coarse);
/* Operator::AddMult with a == 1 (operator.cpp:192-212): y += A x on L-vectors (device pointers). */
int pa_op_apply_add(pa_op *op, const double *x, double *y, void *stream);
/* Operator::Mult (operator.cpp:182-190): y = A x. */
int pa_op_mult(pa_op *op, const double *x, double *y, void *stream);
/* Operator::MultTranspose / AddMultTranspose (fem/libceed/operator.cpp:199-240): y (+)= A^T x.  Trial and test
 * evaluation coincide for every integrator of this library, so A^T is the forward kernel with the coefficient matrices
 * transposed; pa_op_is_symmetric tells whether that differs from A at all. */
int pa_op_mult_transpose(pa_op *op, const double *x, double *y, void *stream);
int pa_op_apply_add_transpose(pa_op *op, const double *x, double *y, void *stream);
int pa_op_is_symmetric(const pa_op *op);
/* SURVEY.md 8(f)-1, ComplexWrapperOperator::Mult (linalg/operator.cpp:98-134) for A = A_r + i A_i: y_r = A_r x_r - A_i x_i,
 * y_i = A_i x_r + A_r x_i in ONE pass over the element data instead of four applies.  Available (pa_op_complex_fused = 1) when both
 * operators are single H(curl) tensor-hexahedron blocks on the same space and geometry in the streaming metric form (isotropic
 * materials, Q1 = 4) -- e.g. op_r from pa_op_add_sub_sum for K - w^2 M and op_i = w C: they then share index arrays and
 * quadrature data and differ by per-element scalar coefficients, so the kernel carries the real and the imaginary part of x in
 * neighbouring lane groups and multiplies by (a_r + i a_i) at the quadrature points.  ess_policy: -1 plain; 0 / 1: with the
 * essential list fused into op_r (pa_op_set_essential), entries read as zero and rows set to 0 / x (the real ParOperator's
 * policy; the imaginary one is DIAG_ZERO, linalg/rap.cpp:450-457).
 * pa_op_complex_fused returns 2 for the dense-table form of the same idea (op_r: curl-curl + mass, op_i: mass / curl-curl / both
 * on the same H(curl) space -- tetrahedra, all straight-sided or all curved; any symmetric materials): the 16 element columns of
 * the matrix-core products carry 8 elements x {real, imaginary} part; meshes mixing straight and curved elements and the essential
 * list too (round 5).
 * Round 5, hexahedra with ANISOTROPIC materials (return value 1 as well): both operators in the packed form (symmetric D of each
 * term at every point, 6 or 12 doubles, Q1 = 4, p <= 3) -- the even lane groups load the real operator's D, the odd ones the
 * imaginary operator's, each applies its D to both parts of the quadrature values and hands over the product that belongs to the
 * other part; every byte of both operators' D read once.
 * Return values 2 and 3 (3: hexahedral form): the FIRST sub-operators of op_r and op_i pair up in the one pass; every further
 * dense sub-operator of either (surface terms: absorbing boundaries, lumped ports, the A2(omega) terms of
 * models/spaceoperator.cpp:786-804) is applied after it to both parts of x, with the signs of operator.cpp:98-134. */
int pa_op_complex_fused(const pa_op *op_r, const pa_op *op_i);
int pa_op_mult_complex(pa_op *op_r, pa_op *op_i, const double *xr, const double *xi, double *yr, double *yi, int ess_policy,
                       void *stream);
/* 1 if y = A x runs on the streaming kernels (single tensor-product block, Q1 = 4, packed q-data): callers choosing between
 * pa_op_mult2 and two pa_op_mult calls prefer the latter then */
int pa_op_streams(const pa_op *op);
/* Affine-element compression of the streaming H(curl) hex kernel (round 6): an element with a constant Jacobian has
 * D(q) = w_q D_e, and a batch of four such elements reads 6 | 7 | 12 numbers per ELEMENT instead of per point (the reference
 * stores per-point data unconditionally, fem/mesh.cpp:146-209; the results agree to the rounding noise of the element's
 * Jacobian, gate 1e-13).  out[0] elements, out[1] affine elements found, out[2] of them in all-affine batches (compressed);
 * first tensor H(curl) sub-operator; all zero when there is none or the form is off (PALACE_AMD_STREAM_AFFINE=0). */
int pa_op_stream_affine(const pa_op *op, int32_t out[3]);
/* The E^T gather of the first dense-table sub-operator (round 6): out[0] = 1 if its E-vector keeps the dofs of an ELEMENT together
 * (chosen at creation by counting the 64-byte sectors the gather would read in either layout; PALACE_AMD_DENSE_ELAYOUT=rows | block
 * overrides), 0 for the rows by dof; out[1] = lanes that share the copies of one dof (1, 2, 4, 8: near the average number of copies;
 * PALACE_AMD_DENSE_GATHER_GROUP overrides).  Both zero when the operator has no dense block. */
int pa_op_dense_gather_form(const pa_op *op, int32_t out[2]);
/* number of dense sub-operators that run in the affine form: every element of the block has a constant Jacobian (straight-sided
 * simplices), so the pre-assembled D of a quadrature point is the D of the first point times w_q / w_0 and the kernel reads 6
 * values per field and element instead of 6 Q (PALACE_AMD_DENSE_AFFINE=0 at creation time keeps the general form) */
int pa_op_dense_affine(const pa_op *op);
/* Fused form of what ParOperator::Mult does around the local apply for square operators
 * (linalg/rap.cpp:207-220: tx = x; tx[ess] = 0; ly = A P tx): after pa_op_set_essential(list of
 * essential L-dofs), pa_op_mult_essential computes y = A (x with the listed entries read as zero)
 * without copying x.  The rows of y at essential dofs are NOT fixed up here (rap.cpp:223-233 does
 * that after P^T). */
/* Which list is fused into the operator's index tables: 0 none, 1 this one, -1 a different one (a second
 * ParOperator with another list must then handle its essential dofs outside the kernels). */
int pa_op_essential_state(const pa_op *op, const int32_t *ess, int32_t n);
int pa_op_set_essential(pa_op *op, const int32_t *ess_ldofs, int32_t n);
/* Multi-rank applies (ParOperator::Mult, rap.cpp:195-234, with the conforming prolongation in flight): the local dofs that
 * take part in the halo exchange are declared once; pa_op_mult_after then computes y = A x where the entries of x on those
 * dofs become valid only when `event` (a hipEvent_t recorded on the exchange stream) has completed -- element batches that
 * touch none of them run first, the others after the event.  Operators without the streaming kernels wait up front. */
int pa_op_set_interface_dofs(pa_op *op, const int32_t *ldofs, int32_t n);
int pa_op_mult_after(pa_op *op, const double *x, double *y, void *stream, void *event);
int pa_op_mult_essential(pa_op *op, const double *x, double *y, void *stream);
/* The same with the row fix-up of rap.cpp:223-233 fused into the E^T kernels when the operator supports it:
 * y[ess] = x[ess] (diag_policy 1, DIAG_ONE) or 0 (DIAG_ZERO).  *handled = 1 if the rows were written, 0 if the
 * caller still has to do it. */
int pa_op_mult_essential_diag(pa_op *op, const double *x, double *y, int diag_policy, void *stream, int *handled);

/* Split vectors: y = A x for a multi-rank apply without L-vector copies.  The local dofs [0, n_true) of the operator are read
 * from x and written to y (the caller's true-dof vectors); the ghosts [n_true, lsize) are read from xg0 or xg1 -- two buffers of
 * lsize - n_true doubles, the one to use is given by the parity of the device-resident counter *sel at the time the kernel runs
 * (sel == NULL: xg0), which is how the peer transport's double-buffered mailboxes are read in place and inside recorded graphs
 * (palace_amd/csrc/comm.hpp) -- and written to yg.  ess_policy >= 0: the essential list of pa_op_set_essential (true dofs) is
 * masked on input and the rows are fixed on output as in pa_op_mult_essential_diag (1: y = x there, 0: y = 0); < 0: plain.
 * What the reference does with three copies around CeedOperatorApplyAdd (linalg/rap.cpp:195-234: tx = x, lx = P tx, ly = A lx,
 * y = P^T ly) when P is a halo exchange.  pa_op_supports_split: 1 for a single H(curl) hexahedral block on the
 * streaming kernels (four or five points per direction: orders 1-4), 0 otherwise (the caller keeps the L-vector path). */
/* One step of a polynomial smoother consumed inside E^T (round 6).  The accumulated form of the Chebyshev recurrence of
 * linalg/chebyshev.cpp:204-218 (e_k = d_0 + ... + d_{k-1}) needs t = A e_k only once, in
 *     out (+)= e_k + sd (e_k - e_prev) + sr dinv .* (r0 - t);
 * pa_op_mult_cheb_step applies the operator to x = e_k with the essential list of pa_op_set_essential fused (rows of t set to
 * x or 0 by diag_policy, as pa_op_mult_essential_diag) and evaluates that line in the epilogue of the E^T gather, which then owns
 * every dof: t is never stored or re-read.  e_prev may be NULL (zero); out may be e_prev's buffer; add != 0: out += ...
 * pa_op_prepare_fused_step builds the index copies this needs, once, outside any stream capture: *available = 0 when the operator
 * has no such form (anything but one tensor H(curl) block on the four-point streaming kernel), and the caller keeps
 * pa_op_mult_essential_diag + its own vector kernel. */
typedef struct {
  double sd, sr;
  const double *dinv, *r0, *e_prev;
  double *out;
  int32_t add;
} pa_cheb_step;
int pa_op_prepare_fused_step(pa_op *op, int *available);
int pa_op_mult_cheb_step(pa_op *op, const double *x, const pa_cheb_step *step, int diag_policy, void *stream);
/* The residual in the same place: res = b - A y (gmg.cpp:186-188; chebyshev.cpp:196-200 with an initial guess) and, when d0 is
 * given, the polynomial's first direction d0 = c0 dinv .* (b - A y) (chebyshev.cpp:201-203) -- A y is not stored.  Either output
 * may be NULL; the essential rows of A y are x | 0 as in pa_op_mult_essential_diag.  Needs pa_op_prepare_fused_step. */
int pa_op_mult_residual(pa_op *op, const double *y, const double *b, double *res, const double *dinv, double c0, double *d0,
                        int diag_policy, void *stream);
/* The same two fused forms on SPLIT vectors (pa_op_mult_split: a multi-rank ParOperator::Mult without L-vector copies).  This rank's
 * gather cannot finish the owned dofs other ranks hold as ghosts (bit 2 of iface_mask[d], d < n_true): their partial sums go to
 * t_iface[d], the ghost rows to yg as in pa_op_mult_split, and the caller's halo kernel (comm.hpp: Halo::RestrictAddDirectStep) adds
 * the neighbours' rows and applies the step there.  mode 1: the Chebyshev step of pa_op_mult_cheb_step; mode 2: the residual of
 * pa_op_mult_residual (res / out as there; sd unused, sr = c0).  ess_policy >= 0: the essential list fused as in pa_op_mult_split. */
typedef struct {
  int32_t mode;
  double sd, sr;
  const double *dinv, *r0, *e_prev;
  double *out;
  int32_t add;
  double *res;
  const uint8_t *iface_mask;
  double *t_iface;
} pa_split_step;
int pa_op_mult_split_step(pa_op *op, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *yg,
                          int n_true, int ess_policy, const pa_split_step *step, void *stream);
int pa_op_supports_split(const pa_op *op);
int pa_op_mult_split(pa_op *op, const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *y,
                     double *yg, int n_true, int ess_policy, void *stream);
/* Two right-hand sides in one pass: y0 = A x0, y1 = A x1.  This is what ComplexWrapperOperator::Mult needs
 * (linalg/operator.cpp:98-134: Ar and Ai are each applied to the real and to the imaginary part); the element's index
 * arrays and D-stage data are read once for both vectors.  The *_essential_diag form is the two-vector version of
 * pa_op_mult_essential_diag. */
int pa_op_mult2(pa_op *op, const double *x0, const double *x1, double *y0, double *y1, void *stream);
int pa_op_mult2_essential_diag(pa_op *op, const double *x0, const double *x1, double *y0, double *y1, int diag_policy,
                               void *stream, int *handled);
/* Operator::AssembleDiagonal (operator.cpp:116-143): diag = diag(A) (zeroed first). */
int pa_op_assemble_diagonal(pa_op *op, double *diag, void *stream);
/* CeedOperatorFullAssemble (operator.cpp:455-523; BilinearForm::FullAssemble / ParOperator::ParallelAssemble,
 * linalg/rap.cpp:84-152): the local operator as CSR in device memory, for coarse solvers that need a
 * matrix.  Pattern = union of the element connectivities (sorted columns); values recovered from the
 * operator's own apply (one apply per colour of a distance-2 colouring), so they are exactly what Mult
 * produces.  skip_zeros drops entries that are exactly zero (operator.cpp:262-313). */
typedef struct pa_csr pa_csr;
int pa_op_full_assemble(pa_op *op, int skip_zeros, void *stream, pa_csr **csr);
/* Device pointers (int32 row pointers / column indices, double values); any output may be NULL. */
int pa_csr_get(const pa_csr *csr, int32_t *nrows, int64_t *nnz, const int32_t **rowptr, const int32_t **colidx,
               const double **values);
/* Columns of an assembled operator: its row count unless it was assembled from a two-space operator (BilinearForm(trial,
 * test)::FullAssemble, e.g. Atn of models/modeeigensolver.cpp:45-56: rows = test dofs, columns = trial dofs). */
int pa_csr_num_cols(const pa_csr *csr);
void pa_csr_destroy(pa_csr *csr);
/* Operator::Size() (operator.hpp:46): number of sub-operators added so far. */
int pa_op_num_sub(const pa_op *op);
/* Operator::DestroyAssemblyData() (operator.cpp:103-114: CeedOperatorAssemblyDataStrip on every sub-operator after
 * BilinearForm::Assemble has built the coarse matrix, bilinearform.cpp:133-141).  This library keeps no assembly state
 * between calls -- pa_op_full_assemble and pa_op_assemble_diagonal release their workspaces before returning -- so the call
 * only validates the handle; it exists so that the reference's call sequence maps one to one. */
int pa_op_destroy_assembly_data(const pa_op *op);
int pa_op_height(const pa_op *op);
int pa_op_width(const pa_op *op);
/* Algorithmic HBM bytes of one apply_add by SURVEY.md 8(d)'s formula
 * NE (Q G 8 + P (4 + o)) + 16 N_L with G = 11, o = 1. */
double pa_op_algorithmic_bytes(const pa_op *op);
void pa_op_destroy(pa_op *op);

#ifdef __cplusplus
}
#endif
#endif /* PALACE_AMD_H */
