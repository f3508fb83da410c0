// Fused E -> B -> D -> B^T -> E^T for NON-tensor elements (tetrahedra, prisms, and any element the
// caller only has dense tables for), gfx950, FP64, on the matrix cores.
//
// This is the path Palace's InitNonTensorBasis feeds (reference fem/libceed/basis.cpp:40-85): the
// basis is a dense [qcomp*Q x P] table per evaluation mode, the restriction is plain, oriented or
// curl-oriented (fem/libceed/restriction.cpp:207-385), and the pointwise D stage is the same set of
// QFunctions as on hexahedra (fem/qfunctions/33/*.h).  For tetrahedra the contraction is a genuine small
// GEMM, [qcomp*Q x P] x [P x elements], so it runs on v_mfma_f64_16x16x4_f64 with 16 ELEMENTS as
// the N dimension of every tile:
//
//   wave      = one block of 16 elements; lane = (kq = lane >> 4, j = lane & 15), j = element
//   B operand = element dofs: lane (kq, j) holds u[dof = 4 s + kq] of element j for step s, i.e. it
//               holds exactly the entries it gathered itself (E writes straight into MFMA layout)
//   A operand = table fragments, pre-swizzled on the host so that one fragment is 64 consecutive
//               doubles (512 B, one coalesced load per MFMA), shared by all waves through L2/L1
//   C result  = row (lane >> 4) + 4 reg, col = element: the rows of one tile are ordered
//               (point group, component) so that a lane ends up with ALL components of its quadrature
//               points (D needs no exchange), and the C layout of the forward product is already the
//               B-operand layout of the transposed product (B^T needs no exchange either)
//   chunks    = 16 quadrature points at a time: forward tiles -> D -> accumulate B^T into P/16 tiles
//   geometry  = element-blocked [block][11][Qpad][16]: each load instruction reads 512 contiguous bytes
//   E^T       = E-vector (coalesced) + the deterministic gather kernel shared with the hex path
//
// No LDS is needed except for the neighbour exchange of the curl-oriented (tridiagonal) restriction.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "pa_device.hpp"
#include "pa_internal.hpp"

namespace pa {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kEB = 16;  // elements per block (N of the MFMA tile)

enum DenseMode {
  MODE_CURL = 0,      // ND curl-curl              f_apply_hdiv_33 on curl u
  MODE_VMASS = 1,     // ND mass                   f_apply_hcurl_33 on u
  MODE_CURLMASS = 2,  // ND curl-curl + mass       f_apply_hdivmass_33
  MODE_DIFF = 3,      // H1 diffusion              f_apply_hcurl_33 on grad u
  MODE_DIFFMASS = 4,  // H1 diffusion + mass       f_apply_hcurlmass_33
  MODE_MASS = 5       // H1 mass                   f_apply_h1_1
};

template <int MODE>
struct ModeTraits {
  static constexpr int NCI = (MODE == MODE_VMASS || MODE == MODE_CURLMASS) ? 3 : (MODE == MODE_DIFFMASS || MODE == MODE_MASS) ? 1 : 0;
  static constexpr int NCD = (MODE == MODE_VMASS || MODE == MODE_MASS) ? 0 : 3;
  static constexpr int NCT = NCI + NCD;
};

int mode_of(int fe_type, int qf) {
  if (fe_type == PA_FE_HCURL) {
    if (qf == PA_QF_HDIV_33) return MODE_CURL;
    if (qf == PA_QF_HCURL_33) return MODE_VMASS;
    if (qf == PA_QF_HDIVMASS_33) return MODE_CURLMASS;
  } else {
    if (qf == PA_QF_HCURL_33) return MODE_DIFF;
    if (qf == PA_QF_HCURLMASS_33) return MODE_DIFFMASS;
    if (qf == PA_QF_H1_1) return MODE_MASS;
  }
  throw Error("QFunction does not match the element type");
}

void mode_comps(int mode, int &nci, int &ncd) {
  nci = (mode == MODE_VMASS || mode == MODE_CURLMASS) ? 3 : (mode == MODE_DIFFMASS || mode == MODE_MASS) ? 1 : 0;
  ncd = (mode == MODE_VMASS || mode == MODE_MASS) ? 0 : 3;
}

struct DenseArgs {
  int ne, nb, P, Q, Qpad, nch, KP;
  const int32_t *idx;
  const uint32_t *co;
  const double *geom;
  const double *Tf, *Tt;
  const double *x;
  double *ye;
  CoeffDev c0, c1;
};

// The pointwise D stage on the NCT components of one quadrature point (in place).
template <int MODE>
__device__ __forceinline__ void dense_D(const DenseArgs &a, const double wdetJ, const double (&adj)[9], const int attr,
                                        double *v) {
  double Cm[9];
  if (MODE == MODE_CURL) {  // hdiv_33_qf.h:10-30
    double Jl[9];
    coeff_unpack3(a.c0, attr, Cm);
    adjJt33(adj, Jl);
    mult_AtBCx33(Jl, Cm, Jl, v[0], v[1], v[2], wdetJ, v[0], v[1], v[2]);
  } else if (MODE == MODE_VMASS || MODE == MODE_DIFF) {  // hcurl_33_qf.h:10-28
    coeff_unpack3(a.c0, attr, Cm);
    mult_AtBCx33(adj, Cm, adj, v[0], v[1], v[2], wdetJ, v[0], v[1], v[2]);
  } else if (MODE == MODE_CURLMASS) {  // hdivmass_33_qf.h:10-44 (mass context first)
    double Jl[9];
    coeff_unpack3(a.c0, attr, Cm);
    mult_AtBCx33(adj, Cm, adj, v[0], v[1], v[2], wdetJ, v[0], v[1], v[2]);
    coeff_unpack3(a.c1, attr, Cm);
    adjJt33(adj, Jl);
    mult_AtBCx33(Jl, Cm, Jl, v[3], v[4], v[5], wdetJ, v[3], v[4], v[5]);
  } else if (MODE == MODE_DIFFMASS) {  // hcurlmass_33_qf.h (scalar mass context first)
    v[0] *= a.c0.mat[coeff_index(a.c0, attr)] * wdetJ;
    coeff_unpack3(a.c1, attr, Cm);
    mult_AtBCx33(adj, Cm, adj, v[1], v[2], v[3], wdetJ, v[1], v[2], v[3]);
  } else {  // h1_1_qf.h
    v[0] *= a.c0.mat[coeff_index(a.c0, attr)] * wdetJ;
  }
}

constexpr int kDenseWaves = 4;

template <int PT, int MODE>
__global__ __launch_bounds__(64 * kDenseWaves, (PT <= 4 ? 2 : 1)) void dense_apply_kernel(const DenseArgs a) {
  using M = ModeTraits<MODE>;
  constexpr int NCT = M::NCT, KPMAX = 4 * PT;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kDenseWaves + wave;
  if (b >= a.nb) return;
  const int j = lane & 15, kq = lane >> 4;
  const int KP = a.KP;

  // ---- E: gather straight into the MFMA B-operand layout
  double u[KPMAX];
  const int32_t *idx = a.idx + (size_t)b * KP * 64;
#pragma unroll
  for (int s = 0; s < KPMAX; s++) {
    u[s] = 0.0;
    if (s < KP) {
      const int sg = idx[s * 64 + lane];
      const int d = sg >= 0 ? sg : -1 - sg;
      const double xv = (d & kEssBit) ? 0.0 : a.x[d & ~kEssBit];
      u[s] = sg >= 0 ? xv : -xv;
    }
  }
  const uint32_t *co = a.co ? a.co + (size_t)b * KP * 64 : nullptr;
  double *sm = smem + (size_t)wave * KPMAX * 64;
  if (co) {  // curl-oriented: u_e = T x_e, T tridiagonal (restriction.cpp:299-369)
#pragma unroll
    for (int s = 0; s < KPMAX; s++)
      if (s < KP) sm[s * 64 + lane] = u[s];
    wave_sync();
#pragma unroll
    for (int s = 0; s < KPMAX; s++) {
      if (s < KP) {
        const uint32_t c = co[s * 64 + lane];
        const int dof = 4 * s + kq;
        const double lo = dof > 0 ? sm[(dof - 1) * 16 + j] : 0.0;
        const double hi = dof + 1 < 4 * KP ? sm[(dof + 1) * 16 + j] : 0.0;
        u[s] = (double)(int8_t)(c & 0xff) * lo + (double)(int8_t)((c >> 8) & 0xff) * u[s] +
               (double)(int8_t)((c >> 16) & 0xff) * hi;
      }
    }
    wave_sync();
  }

  double4_t yacc[PT];
#pragma unroll
  for (int pt = 0; pt < PT; pt++) yacc[pt] = double4_t{0.0, 0.0, 0.0, 0.0};

  for (int c = 0; c < a.nch; c++) {
    // geometry of this lane's 4 points of the chunk (q = 16 c + 4 gl + kq), requested up front
    double gd[4][10];
    int attr[4];
    {
      const double *g = a.geom + ((size_t)b * 11 * a.Qpad + 16 * c) * kEB + lane;
#pragma unroll
      for (int gl = 0; gl < 4; gl++) {
        attr[gl] = (a.c0.nattr > 0 || a.c1.nattr > 0) ? max(1, (int)g[gl * 64]) : 1;
#pragma unroll
        for (int k = 0; k < 10; k++) gd[gl][k] = g[((size_t)(1 + k) * a.Qpad) * kEB + gl * 64];
      }
    }
    // ---- B: NCT tiles of 16 rows = 4 points x (4 NCT) (group, component) pairs
    double4_t acc[NCT];
#pragma unroll
    for (int t = 0; t < NCT; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double *tf = a.Tf + (size_t)c * KP * NCT * 64 + lane;
#pragma unroll
    for (int s = 0; s < KPMAX; s++) {
      if (s < KP) {
#pragma unroll
        for (int t = 0; t < NCT; t++)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tf[(s * NCT + t) * 64], u[s], acc[t], 0, 0, 0);
      }
    }
    // ---- D
#pragma unroll
    for (int gl = 0; gl < 4; gl++) {
      double v[NCT];
#pragma unroll
      for (int k = 0; k < NCT; k++) v[k] = acc[(gl * NCT + k) >> 2][(gl * NCT + k) & 3];
      double adj[9];
#pragma unroll
      for (int k = 0; k < 9; k++) adj[k] = gd[gl][1 + k];
      dense_D<MODE>(a, gd[gl][0], adj, attr[gl], v);
#pragma unroll
      for (int k = 0; k < NCT; k++) acc[(gl * NCT + k) >> 2][(gl * NCT + k) & 3] = v[k];
    }
    // ---- B^T: the C layout above is the B-operand layout of the transposed product
    const double *tt = a.Tt + (size_t)c * 4 * NCT * PT * 64 + lane;
#pragma unroll
    for (int pi = 0; pi < 4 * NCT; pi++) {
#pragma unroll
      for (int pt = 0; pt < PT; pt++)
        yacc[pt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tt[(pi * PT + pt) * 64], acc[pi >> 2][pi & 3], yacc[pt], 0, 0, 0);
    }
  }

  // ---- E^T, first half: E-vector in the same [dof][element] block layout (coalesced); signs of the
  // oriented restriction and the sum over elements happen in the gather kernel
  double *ye = a.ye + (size_t)b * KP * 64;
  if (co) {  // w = T^T y_e
#pragma unroll
    for (int s = 0; s < KPMAX; s++)
      if (s < KP) sm[s * 64 + lane] = yacc[s >> 2][s & 3];
    wave_sync();
#pragma unroll
    for (int s = 0; s < KPMAX; s++) {
      if (s < KP) {
        const int dof = 4 * s + kq;
        const uint32_t cm = co[s * 64 + lane];
        double w = (double)(int8_t)((cm >> 8) & 0xff) * yacc[s >> 2][s & 3];
        if (dof > 0) w += (double)(int8_t)((co[s * 64 + lane - 16] >> 16) & 0xff) * sm[(dof - 1) * 16 + j];
        if (dof + 1 < 4 * KP) w += (double)(int8_t)(co[s * 64 + lane + 16] & 0xff) * sm[(dof + 1) * 16 + j];
        ye[s * 64 + lane] = w;
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KPMAX; s++)
      if (s < KP) ye[s * 64 + lane] = yacc[s >> 2][s & 3];
  }
}

template <int PT>
void launch_pt(const DenseSub &ds, const DenseArgs &a, hipStream_t s) {
  const dim3 grid((ds.nb + kDenseWaves - 1) / kDenseWaves), block(64 * kDenseWaves);
  const size_t shm = ds.d_co ? sizeof(double) * kDenseWaves * 4 * PT * 64 : 0;
  switch (ds.mode) {
#define PA_DENSE_CASE(MODE) \
  case MODE: hipLaunchKernelGGL((dense_apply_kernel<PT, MODE>), grid, block, shm, s, a); break;
    PA_DENSE_CASE(MODE_CURL)
    PA_DENSE_CASE(MODE_VMASS)
    PA_DENSE_CASE(MODE_CURLMASS)
    PA_DENSE_CASE(MODE_DIFF)
    PA_DENSE_CASE(MODE_DIFFMASS)
    PA_DENSE_CASE(MODE_MASS)
#undef PA_DENSE_CASE
  }
}

// ---- geometry factors, element-blocked (set-up; one thread per point) ---------------------------
__global__ void geom_dense_kernel(const int ne, const int Q, const int Qpad, const int npe,
                                  const int32_t *__restrict__ node_off, const double *__restrict__ nodes,
                                  const int32_t *__restrict__ attr, const double *__restrict__ grad,
                                  const double *__restrict__ w, double *__restrict__ geom) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int n = 0; n < npe; n++) {
    const int id = node_off[(size_t)e * npe + n];
    const double d0 = grad[((size_t)0 * Q + q) * npe + n], d1 = grad[((size_t)1 * Q + q) * npe + n],
                 d2 = grad[((size_t)2 * Q + q) * npe + n];
    for (int c = 0; c < 3; c++) {
      const double X = nodes[3 * (size_t)id + c];
      J[c + 0] += X * d0;
      J[c + 3] += X * d1;
      J[c + 6] += X * d2;
    }
  }
  // fem/qfunctions/33/geom_33_qf.h:9-33
  double A[9];
  adjJt33(J, A);
  const double det = J[0] * A[0] + J[1] * A[1] + J[2] * A[2];
  double *g = geom + ((size_t)(e / kEB) * 11 * Qpad + q) * kEB + (e % kEB);
  const size_t cs = (size_t)Qpad * kEB;
  g[0] = (double)attr[e];
  g[cs] = w[q] * det;
  for (int c = 0; c < 9; c++) g[(2 + c) * cs] = A[c] / det;
}

// ---- diagonal (set-up): one thread per (element, local dof) --------------------------------------
// CeedOperatorLinearAssembleAddDiagonal [libCEED, external]: element diagonals d_e[j] = sum_q b_j^T D b_j
// pushed through the transpose of the UNSIGNED restriction (for the curl-oriented one: |T|^T d_e).
template <int MODE>
__global__ void dense_diag_kernel(const DenseArgs a, const int32_t *__restrict__ off, const int8_t *__restrict__ cor,
                                  const double *__restrict__ interp, const double *__restrict__ deriv,
                                  double *__restrict__ diag) {
  using M = ModeTraits<MODE>;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / a.P);
  if (e >= a.ne) return;
  const int jd = (int)(gid - (long long)e * a.P);
  const size_t cs = (size_t)a.Qpad * kEB;
  const double *g = a.geom + ((size_t)(e / kEB) * 11 * a.Qpad) * kEB + (e % kEB);
  double d = 0.0;
  for (int q = 0; q < a.Q; q++) {
    double v[M::NCT > 0 ? M::NCT : 1], w[M::NCT > 0 ? M::NCT : 1];
    for (int k = 0; k < M::NCI; k++) v[k] = interp[((size_t)k * a.Q + q) * a.P + jd];
    for (int k = 0; k < M::NCD; k++) v[M::NCI + k] = deriv[((size_t)k * a.Q + q) * a.P + jd];
    for (int k = 0; k < M::NCT; k++) w[k] = v[k];
    double adj[9];
    for (int k = 0; k < 9; k++) adj[k] = g[(2 + k) * cs + (size_t)q * kEB];
    dense_D<MODE>(a, g[cs + (size_t)q * kEB], adj, (int)g[(size_t)q * kEB], w);
    for (int k = 0; k < M::NCT; k++) d += v[k] * w[k];
  }
  const int32_t *oe = off + (size_t)e * a.P;
  if (cor) {
    const int8_t *t = cor + 3 * ((size_t)e * a.P + jd);
    if (t[1]) unsafeAtomicAdd(&diag[oe[jd]], fabs((double)t[1]) * d);
    if (jd > 0 && t[0]) unsafeAtomicAdd(&diag[oe[jd - 1]], fabs((double)t[0]) * d);
    if (jd + 1 < a.P && t[2]) unsafeAtomicAdd(&diag[oe[jd + 1]], fabs((double)t[2]) * d);
  } else {
    unsafeAtomicAdd(&diag[oe[jd]], d);
  }
}

DenseArgs make_args(const DenseSub &ds) {
  DenseArgs a;
  a.ne = ds.ne, a.nb = ds.nb, a.P = ds.P, a.Q = ds.Q, a.Qpad = ds.Qpad, a.nch = ds.nch, a.KP = ds.KP;
  a.idx = ds.d_idx, a.co = ds.d_co, a.geom = ds.geom->d_geom, a.Tf = ds.d_Tf, a.Tt = ds.d_Tt;
  a.x = nullptr, a.ye = ds.d_ye;
  a.c0 = ds.c0.dev(), a.c1 = ds.c1.dev();
  return a;
}

}  // namespace

void launch_geom_dense(const pa_mesh_dense_desc &mesh, Geom &g, hipStream_t s) {
  const int ne = mesh.num_elem, npe = mesh.nodes_per_elem, Q = mesh.num_qpts;
  PA_REQUIRE(ne > 0 && npe > 0 && Q > 0 && mesh.num_nodes > 0, "empty mesh description");
  PA_REQUIRE(mesh.node_offsets && mesh.nodes && mesh.attr && mesh.mesh_grad && mesh.qweight, "null mesh array");
  for (size_t i = 0; i < (size_t)ne * npe; i++)
    PA_REQUIRE(mesh.node_offsets[i] >= 0 && mesh.node_offsets[i] < mesh.num_nodes, "mesh node id out of range");
  for (int e = 0; e < ne; e++) PA_REQUIRE(mesh.attr[e] >= 1, "element attributes are 1-based");
  int32_t *d_off = dev_upload(mesh.node_offsets, (size_t)ne * npe, s);
  double *d_nodes = dev_upload(mesh.nodes, (size_t)mesh.num_nodes * 3, s);
  int32_t *d_attr = dev_upload(mesh.attr, (size_t)ne, s);
  double *d_grad = dev_upload(mesh.mesh_grad, (size_t)3 * Q * npe, s);
  double *d_w = dev_upload(mesh.qweight, (size_t)Q, s);
  g.ne = ne, g.q1d = 0, g.Q = Q, g.eb = kEB, g.Qpad = (Q + 15) / 16 * 16;
  const size_t nb = (size_t)(ne + kEB - 1) / kEB, count = nb * 11 * g.Qpad * kEB;
  g.d_geom = dev_alloc<double>(count);
  PA_HIP(hipMemsetAsync(g.d_geom, 0, sizeof(double) * count, s));
  const long long n = (long long)ne * Q;
  const int bs = 256;
  hipLaunchKernelGGL(geom_dense_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                     d_nodes, d_attr, d_grad, d_w, g.d_geom);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  hipFree(d_off), hipFree(d_nodes), hipFree(d_attr), hipFree(d_grad), hipFree(d_w);
}

DenseSub *make_dense_sub(pa_geom *geom, const pa_restriction_desc &r, const pa_dense_basis_desc &b, int qf,
                         const void *ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops, int height) {
  PA_REQUIRE(geom && geom->eb == kEB, "geometry data must come from pa_geom_create_dense");
  PA_REQUIRE(b.fe_type == PA_FE_H1 || b.fe_type == PA_FE_HCURL, "unknown element type");
  PA_REQUIRE(b.num_dofs > 0 && b.num_qpts == geom->Q, "basis and geometry data disagree on the quadrature rule");
  PA_REQUIRE(r.num_elem == geom->ne && r.elem_size == b.num_dofs, "restriction does not match mesh / basis");
  PA_REQUIRE(r.lsize == height && r.offsets, "restriction L-vector size does not match the operator");
  PA_REQUIRE(!(r.orients && r.curl_orients), "restriction is either oriented or curl-oriented");
  PA_REQUIRE(r.lsize < (1 << 29), "too many local dofs for the index encoding");
  const int P = b.num_dofs, Q = b.num_qpts, ne = r.num_elem;
  const int mode = mode_of(b.fe_type, qf);
  int nci, ncd;
  mode_comps(mode, nci, ncd);
  const int nct = nci + ncd;
  PA_REQUIRE(nci == 0 || b.interp, "interp table missing");
  PA_REQUIRE(ncd == 0 || b.deriv, "curl / gradient table missing");
  {
    const uint32_t want_i = nci ? PA_EVAL_INTERP : 0u;
    const uint32_t want_d = ncd ? (b.fe_type == PA_FE_HCURL ? PA_EVAL_CURL : PA_EVAL_GRAD) : 0u;
    PA_REQUIRE(trial_ops == (want_i | want_d) && test_ops == trial_ops, "eval modes do not match the QFunction");
  }
  static const int kPT[] = {1, 2, 3, 4, 6, 9};
  int PT = 0;
  for (int v : kPT)
    if (P <= 16 * v) {
      PT = v;
      break;
    }
  PA_REQUIRE(PT > 0, "element has more than 144 dofs: not instantiated");

  auto *ds = new DenseSub;
  ds->geom = geom;
  geom->refcount++;
  ds->fe_type = b.fe_type, ds->P = P, ds->Q = Q, ds->Qpad = geom->Qpad, ds->nch = geom->Qpad / 16;
  ds->ne = ne, ds->nb = (ne + kEB - 1) / kEB, ds->lsize = r.lsize, ds->KP = (P + 3) / 4, ds->PT = PT;
  ds->qf = qf, ds->mode = mode, ds->trial_ops = trial_ops, ds->test_ops = test_ops;
  const int KP = ds->KP, nb = ds->nb, nch = ds->nch;

  // ---- E: block-transposed index (+ packed tridiagonal rows)
  const size_t nslot = (size_t)nb * KP * 64;
  std::vector<int32_t> idx(nslot, kEssBit);
  std::vector<uint32_t> co(r.curl_orients ? nslot : 0, 0u);
  for (int e = 0; e < ne; e++) {
    for (int d = 0; d < P; d++) {
      const int32_t off = r.offsets[(size_t)e * P + d];
      PA_REQUIRE(off >= 0 && off < r.lsize, "restriction offset out of range");
      const size_t pos = ((size_t)(e / kEB) * 4 * KP + d) * kEB + (e % kEB);
      idx[pos] = (r.orients && r.orients[(size_t)e * P + d]) ? -1 - off : off;
      if (r.curl_orients) {
        const int8_t *t = r.curl_orients + 3 * ((size_t)e * P + d);
        co[pos] = (uint32_t)(uint8_t)t[0] | ((uint32_t)(uint8_t)t[1] << 8) | ((uint32_t)(uint8_t)t[2] << 16);
      }
    }
  }
  ds->d_idx = dev_upload(idx.data(), nslot);
  if (r.curl_orients) ds->d_co = dev_upload(co.data(), nslot);
  // transpose map for the gather form of E^T (counting sort by dof, element order preserved)
  {
    std::vector<int32_t> tptr((size_t)r.lsize + 1, 0), tent((size_t)ne * P);
    for (size_t k = 0; k < (size_t)ne * P; k++) tptr[(size_t)r.offsets[k] + 1]++;
    for (int d = 0; d < r.lsize; d++) tptr[d + 1] += tptr[d];
    std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
    for (int e = 0; e < ne; e++)
      for (int d = 0; d < P; d++) {
        const int32_t off = r.offsets[(size_t)e * P + d];
        const int32_t pos = (int32_t)(((size_t)(e / kEB) * 4 * KP + d) * kEB + (e % kEB));
        const bool flip = r.orients && r.orients[(size_t)e * P + d];
        tent[fill[off]++] = flip ? -1 - pos : pos;
      }
    ds->d_tptr = dev_upload(tptr.data(), tptr.size());
    ds->d_tent = dev_upload(tent.data(), tent.size());
    ds->d_ye = dev_alloc<double>(nslot);
  }
  ds->h_idx = std::move(idx);

  // ---- B: MFMA A-operand fragments (see the header comment for the row order)
  auto tab = [&](int cidx, int q, int p) -> double {
    if (q >= Q || p >= P) return 0.0;
    return cidx < nci ? b.interp[((size_t)cidx * Q + q) * P + p] : b.deriv[((size_t)(cidx - nci) * Q + q) * P + p];
  };
  std::vector<double> Tf((size_t)nch * KP * nct * 64), Tt((size_t)nch * 4 * nct * PT * 64);
  for (int c = 0; c < nch; c++) {
    for (int s = 0; s < KP; s++)
      for (int t = 0; t < nct; t++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, kq = lane >> 4;
          const int pi = 4 * t + (i >> 2), gl = pi / nct, cidx = pi % nct;
          Tf[(((size_t)c * KP + s) * nct + t) * 64 + lane] = tab(cidx, 16 * c + 4 * gl + (i & 3), 4 * s + kq);
        }
    for (int pi = 0; pi < 4 * nct; pi++)
      for (int pt = 0; pt < PT; pt++)
        for (int lane = 0; lane < 64; lane++) {
          const int i = lane & 15, kq = lane >> 4;
          const int gl = pi / nct, cidx = pi % nct;
          Tt[(((size_t)c * 4 * nct + pi) * PT + pt) * 64 + lane] = tab(cidx, 16 * c + 4 * gl + kq, 16 * pt + i);
        }
  }
  ds->d_Tf = dev_upload(Tf.data(), Tf.size());
  ds->d_Tt = dev_upload(Tt.data(), Tt.size());
  // plain copies for the diagonal kernel
  if (nci) ds->d_interp = dev_upload(b.interp, (size_t)nci * Q * P);
  if (ncd) ds->d_deriv = dev_upload(b.deriv, (size_t)3 * Q * P);
  ds->d_off = dev_upload(r.offsets, (size_t)ne * P);
  if (r.curl_orients) ds->d_cor = dev_upload(r.curl_orients, (size_t)3 * ne * P);

  // ---- D: coefficient context(s)
  PA_REQUIRE(ctx && ctx_size >= 16 && ctx_size % 8 == 0, "bad coefficient context");
  ds->ctx_blob.assign((const uint8_t *)ctx, (const uint8_t *)ctx + ctx_size);
  switch (mode) {
    case MODE_CURL:
    case MODE_VMASS:
    case MODE_DIFF:
      parse_coeff(ctx, ctx_size, 3, ds->c0, 0);
      break;
    case MODE_CURLMASS:
      parse_coeff(ctx, ctx_size, 3, ds->c0, 0);
      parse_coeff(ctx, ctx_size, 3, ds->c1, ds->c0.slots);
      break;
    case MODE_DIFFMASS:
      parse_coeff(ctx, ctx_size, 1, ds->c0, 0);
      parse_coeff(ctx, ctx_size, 3, ds->c1, ds->c0.slots);
      break;
    case MODE_MASS:
      parse_coeff(ctx, ctx_size, 1, ds->c0, 0);
      break;
  }
  return ds;
}

void free_dense_sub(DenseSub *ds) {
  if (!ds) return;
  hipFree(ds->d_idx), hipFree(ds->d_idx_bc), hipFree(ds->d_co);
  hipFree(ds->d_Tf), hipFree(ds->d_Tt), hipFree(ds->d_interp), hipFree(ds->d_deriv);
  hipFree(ds->d_off), hipFree(ds->d_cor), hipFree(ds->d_ori);
  hipFree(ds->d_ye), hipFree(ds->d_tptr), hipFree(ds->d_tent);
  hipFree(ds->c0.d_attr_mat), hipFree(ds->c0.d_mat);
  hipFree(ds->c1.d_attr_mat), hipFree(ds->c1.d_mat);
  pa_geom_destroy(static_cast<pa_geom *>(ds->geom));
  delete ds;
}

void dense_set_essential(DenseSub &ds, const std::vector<char> &flag) {
  std::vector<int32_t> bc(ds.h_idx);
  for (auto &s : bc) {
    const int d = s >= 0 ? s : -1 - s;
    if (d & kEssBit) continue;  // padding
    if (flag[d]) s = s >= 0 ? (d | kEssBit) : -1 - (d | kEssBit);
  }
  hipFree(ds.d_idx_bc);
  ds.d_idx_bc = dev_upload(bc.data(), bc.size());
}

void launch_dense_apply(const DenseSub &ds, const double *x, bool masked, hipStream_t s) {
  DenseArgs a = make_args(ds);
  a.x = x;
  if (masked && ds.d_idx_bc) a.idx = ds.d_idx_bc;
  switch (ds.PT) {
    case 1: launch_pt<1>(ds, a, s); break;
    case 2: launch_pt<2>(ds, a, s); break;
    case 3: launch_pt<3>(ds, a, s); break;
    case 4: launch_pt<4>(ds, a, s); break;
    case 6: launch_pt<6>(ds, a, s); break;
    case 9: launch_pt<9>(ds, a, s); break;
    default: throw Error("dense kernel not instantiated for this element size");
  }
  PA_HIP(hipGetLastError());
}

void launch_dense_gather(const DenseSub &ds, double *y, bool accumulate, hipStream_t s) {
  launch_et_gather_raw(ds.lsize, ds.d_tptr, ds.d_tent, ds.d_ye, y, accumulate, s);
}

void launch_dense_diag(const DenseSub &ds, double *diag, hipStream_t s) {
  DenseArgs a = make_args(ds);
  const long long n = (long long)ds.ne * ds.P;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (ds.mode) {
#define PA_DIAG_CASE(MODE)                                                                                   \
  case MODE:                                                                                                 \
    hipLaunchKernelGGL((dense_diag_kernel<MODE>), grid, block, 0, s, a, ds.d_off, ds.d_cor, ds.d_interp, ds.d_deriv, \
                       diag);                                                                                \
    break;
    PA_DIAG_CASE(MODE_CURL)
    PA_DIAG_CASE(MODE_VMASS)
    PA_DIAG_CASE(MODE_CURLMASS)
    PA_DIAG_CASE(MODE_DIFF)
    PA_DIAG_CASE(MODE_DIFFMASS)
    PA_DIAG_CASE(MODE_MASS)
#undef PA_DIAG_CASE
  }
  PA_HIP(hipGetLastError());
}

}  // namespace pa
