// Operators between two different spaces on the same elements, and the element-wise error integrators built on them: the
// pieces of the flux error estimators (linalg/errorestimator.cpp) that are not same-space bilinear forms.
//
//  * mixed mass  (v, C u)  with u in H(curl) and v in H(div) or the other way round: FluxProjector's `Flux` operator
//    (errorestimator.cpp:164-176, BilinearForm(rhs_fespace, smooth_fespace) + VectorFEMassIntegrator, which picks
//    f_apply_hcurlhdiv_33 / f_apply_hdivhcurl_33 by the map types of the two elements, fem/integ/vecfemass.cpp:88-101);
//  * element error  eta_e^2 += int_e |C_2 u_2 - C_1 u_1|^2  for u_1, u_2 in those two spaces
//    (AssembleCeedElementErrorIntegrator, fem/libceed/integrator.cpp:550-626, with f_apply_hcurlhdiv_error_33 /
//    f_apply_hdivhcurl_error_33, fem/qfunctions/33/hcurlhdiv_error_33_qf.h:10-78).
//
//  * the same-map mixed form  (C grad phi, v)  with phi in H1 and v in H(curl): MixedVectorGradientIntegrator
//    (fem/integ/mixedvecgrad.cpp:43-76, f_apply_hcurl_33 / f_apply_hcurl_22 with trial Grad and test Interp) -- the `Atn` block
//    of the boundary-mode eigenproblem (models/modeeigensolver.cpp:52); an H1 side enters with its gradient table;
//  * the scalar pair of the 2-D curl flux estimator (errorestimator.cpp:122-160, :446-474: the curl of a plane field is a
//    scalar): mass (c u, v) between two scalar spaces (MassIntegrator, f_apply_h1_1) and the element error
//    f_apply_l2h1_error (fem/qfunctions/l2h1_error_qf.h:14-30), scalar coefficients, only w detJ of the geometry data;
//  * all of them on triangles / quadrilaterals in the plane (qfunctions/22/hcurlhdiv_22_qf.h, hcurlhdiv_error_22_qf.h): the
//    2 x 2 Jacobian adjugate and coefficient are embedded in 3 x 3 matrices (third component of every field zero), which
//    makes the 3-D arithmetic below the reference's 2-D arithmetic term by term.
//
//  * the members of those families on boundary and line elements (qfunctions/32 | 31 | 21: hcurlhdiv_*, hcurl_* between two
//    spaces) and the gradient form  (C grad u, v)  with v in a vector H1 space (GradientIntegrator, fem/integ/grad.cpp:16-72,
//    f_apply_hcurlh1d_* on every geometry): mixed_embedded_kernel below.  No driver of the reference adds them.
//
// These run once per solve (post-processing), not inside the Krylov loop: one wave per element, dense tables read through the
// caches (q fastest for the forward product, dof fastest for the transposed one: coalesced either way), the pointwise arithmetic
// exactly the reference QFunctions' on the element-blocked geometry data of pa_geom_create_dense, E^T as E-vector + the
// deterministic gather.  Any element type the caller has dense tables for (tetrahedra and hexahedra are tested).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "pa_device.hpp"

namespace pa {

namespace {

constexpr int kMixWaves = 4;
constexpr int kEBm = 16;  // element block of the dense geometry layout

struct MixSideDev {
  int P, nc;  // dofs per element, components of the value table (2 in the plane)
  const int32_t *sidx;  // [ne][P] L-vector index, -(1 + index) when the sign is flipped
  const int8_t *cor;    // [ne][P][3] rows of the tridiagonal dof transformation or nullptr
  const double *tabF;   // [nc][P][Q] values at the quadrature points, point fastest
  const double *tabT;   // [nc][Q][P] the same, dof fastest
};

struct MixArgs {
  int ne, Q, Qpad, stride, dim;
  const double *geom;
  MixSideDev s1, s2;
  CoeffDev c0, c1;
  const double *x1, *x2;
  double *ye;   // [ne][P2] (apply)
  double *out;  // [ne] (error)
};

// u_e = T_e (signed gather of x): restriction.cpp:299-369 for the curl-oriented case
__device__ __forceinline__ void mix_gather(const MixSideDev &sd, const int e, const int lane, const double *__restrict__ x,
                                           double *dst, double *tmp) {
  double *raw = sd.cor ? tmp : dst;
  for (int d = lane; d < sd.P; d += 64) {
    const int sg = sd.sidx[(size_t)e * sd.P + d];
    const double v = x[sg >= 0 ? sg : -1 - sg];
    raw[d] = sg >= 0 ? v : -v;
  }
  wave_sync();
  if (sd.cor) {
    for (int d = lane; d < sd.P; d += 64) {
      const int8_t *t = sd.cor + 3 * ((size_t)e * sd.P + d);
      const double lo = d > 0 ? tmp[d - 1] : 0.0, hi = d + 1 < sd.P ? tmp[d + 1] : 0.0;
      dst[d] = (double)t[0] * lo + (double)t[1] * tmp[d] + (double)t[2] * hi;
    }
    wave_sync();
  }
}

__device__ __forceinline__ void mix_eval(const MixSideDev &sd, const int Q, const int q, const double *xs, double (&u)[3]) {
  u[0] = u[1] = u[2] = 0.0;
  if (sd.nc == 1) {
    for (int d = 0; d < sd.P; d++) u[0] += sd.tabF[(size_t)d * Q + q] * xs[d];
  } else if (sd.nc == 3) {
    for (int d = 0; d < sd.P; d++) {
      const double xv = xs[d];
      u[0] += sd.tabF[((size_t)0 * sd.P + d) * Q + q] * xv;
      u[1] += sd.tabF[((size_t)1 * sd.P + d) * Q + q] * xv;
      u[2] += sd.tabF[((size_t)2 * sd.P + d) * Q + q] * xv;
    }
  } else {
    for (int d = 0; d < sd.P; d++) {
      const double xv = xs[d];
      u[0] += sd.tabF[((size_t)0 * sd.P + d) * Q + q] * xv;
      u[1] += sd.tabF[((size_t)1 * sd.P + d) * Q + q] * xv;
    }
  }
}

// y = B (A x), column-major 3x3 (utils_33_qf.h:86-101)
__device__ __forceinline__ void mult_BAx33(const double A[9], const double B[9], const double (&x)[3], double (&y)[3]) {
  const double z0 = A[0] * x[0] + A[3] * x[1] + A[6] * x[2];
  const double z1 = A[1] * x[0] + A[4] * x[1] + A[7] * x[2];
  const double z2 = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
  y[0] = B[0] * z0 + B[3] * z1 + B[6] * z2;
  y[1] = B[1] * z0 + B[4] * z1 + B[7] * z2;
  y[2] = B[2] * z0 + B[5] * z1 + B[8] * z2;
}

// A 2 x 2 column-major matrix as the leading block of a 3 x 3 one
__device__ __forceinline__ void embed22(double m0, double m1, double m2, double m3, double corner, double M[9]) {
  M[0] = m0, M[1] = m1, M[2] = 0.0, M[3] = m2, M[4] = m3, M[5] = 0.0, M[6] = 0.0, M[7] = 0.0, M[8] = corner;
}
// CoeffUnpack2 (coeff/coeff_2_qf.h) into that form
__device__ __forceinline__ void coeff_unpack2in3(const CoeffDev &c, int attr, double C[9]) {
  const double *m = c.mat + 4 * coeff_index(c, attr);
  embed22(m[0], m[1], m[2], m[3], 0.0, C);
}

// (KIND 8: f_apply_hdiv_33 between two spaces -- MixedVectorCurlIntegrator with an H(div) test space, an H(curl) side entering with
// its curl table, mixedveccurl.cpp:41-46; 7 is the gradient form of mixed_embedded_kernel)
// KIND 0: f_apply_hcurlhdiv_33 | _22, 1: f_apply_hdivhcurl_33 | _22, 2: f_apply_hcurlhdiv_error_33 | _22,
// 3: f_apply_hdivhcurl_error_33 | _22, 4: f_apply_hcurl_33 | _22 between two spaces, 5: f_apply_h1_1 between two scalar
// spaces, 6: f_apply_l2h1_error
template <int KIND>
__global__ __launch_bounds__(64 * kMixWaves) void mixed_kernel(const MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * kMixWaves + wave;
  if (e >= a.ne) return;  // no workgroup barriers below
  constexpr bool ERR = KIND == 2 || KIND == 3 || KIND == 6;
  constexpr bool SCALAR = KIND == 5 || KIND == 6;
  const int P1 = a.s1.P, P2 = a.s2.P, Q = a.Q;
  double *xa = smem + (size_t)wave * a.stride;
  double *xb = xa + P1;
  double *tmp = xb + P2;
  double *vq = tmp + max(P1, P2);

  mix_gather(a.s1, e, lane, a.x1, xa, tmp);
  if (ERR) mix_gather(a.s2, e, lane, a.x2, xb, tmp);

  const int nrows = a.dim == 3 ? 11 : 6;
  const double *g = a.geom + ((size_t)(e / kEBm) * nrows * a.Qpad) * kEBm + (e % kEBm);
  double err = 0.0;
  for (int q = lane; q < Q; q += 64) {
    double u1[3];
    mix_eval(a.s1, Q, q, xa, u1);
    const int attr = (a.c0.nattr > 0 || a.c1.nattr > 0) ? max(1, (int)g[(size_t)q * kEBm]) : 1;
    const double wdetJ = g[((size_t)a.Qpad + q) * kEBm];
    if (SCALAR) {
      const double c1 = a.c0.mat[coeff_index(a.c0, attr)];
      if (KIND == 5) {  // h1_1_qf.h:10-24
        vq[q] = c1 * wdetJ * u1[0];
      } else {  // l2h1_error_qf.h:14-30
        double u2[3];
        mix_eval(a.s2, Q, q, xb, u2);
        const double diff = c1 * u1[0] - a.c1.mat[coeff_index(a.c1, attr)] * u2[0];
        err += wdetJ * diff * diff;
      }
      continue;
    }
    double adj[9], Jl[9], Cm[9];
    if (a.dim == 3) {
#pragma unroll
      for (int k = 0; k < 9; k++) adj[k] = g[((size_t)(2 + k) * a.Qpad + q) * kEBm];
      coeff_unpack3(a.c0, attr, Cm);
    } else {
      embed22(g[((size_t)2 * a.Qpad + q) * kEBm], g[((size_t)3 * a.Qpad + q) * kEBm], g[((size_t)4 * a.Qpad + q) * kEBm],
              g[((size_t)5 * a.Qpad + q) * kEBm], 1.0, adj);
      coeff_unpack2in3(a.c0, attr, Cm);
    }
    adjJt33(adj, Jl);  // in the plane: AdjJt22 (utils_22_qf.h:19-29) in the leading 2 x 2 block
    if (!ERR) {
      double v0, v1, v2;
      if (KIND == 0)  // hcurlhdiv_33_qf.h:10-31: H(curl) trial (adj(J)^T / det J), H(div) test (J / det J)
        mult_AtBCx33(Jl, Cm, adj, u1[0], u1[1], u1[2], wdetJ, v0, v1, v2);
      else if (KIND == 1)  // hcurlhdiv_33_qf.h:33-54
        mult_AtBCx33(adj, Cm, Jl, u1[0], u1[1], u1[2], wdetJ, v0, v1, v2);
      else if (KIND == 8)  // hdiv_33_qf.h:10-30 between two spaces: both sides contravariant (curls of H(curl), values of H(div))
        mult_AtBCx33(Jl, Cm, Jl, u1[0], u1[1], u1[2], wdetJ, v0, v1, v2);
      else  // hcurl_33_qf.h:10-28
        mult_AtBCx33(adj, Cm, adj, u1[0], u1[1], u1[2], wdetJ, v0, v1, v2);
      vq[q] = v0, vq[Q + q] = v1;
      if (a.dim == 3) vq[2 * Q + q] = v2;
    } else {
      double u2[3], w1[3], w2[3], C2[9];
      mix_eval(a.s2, Q, q, xb, u2);
      if (a.dim == 3)
        coeff_unpack3(a.c1, attr, C2);
      else
        coeff_unpack2in3(a.c1, attr, C2);
      if (KIND == 2) {  // hcurlhdiv_error_33_qf.h:10-43
        mult_BAx33(adj, Cm, u1, w1);
        mult_BAx33(Jl, C2, u2, w2);
      } else {  // :45-78
        mult_BAx33(Jl, Cm, u1, w1);
        mult_BAx33(adj, C2, u2, w2);
      }
      w2[0] -= w1[0], w2[1] -= w1[1], w2[2] -= w1[2];
      err += wdetJ * (w2[0] * w2[0] + w2[1] * w2[1] + w2[2] * w2[2]);
    }
  }
  if (ERR) {
    // sum over the points of the element (the all-ones basis of integrator.cpp:560-574), fixed tree
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) err += __shfl_xor(err, m, 64);
    if (lane == 0) a.out[e] += err;
    return;
  }
  wave_sync();
  // B^T on the test side, then T_e^T and the E-vector
  double *yt = tmp;
  for (int i = lane; i < P2; i += 64) {
    double s = 0.0;
    for (int c = 0; c < a.s2.nc; c++)
      for (int q = 0; q < Q; q++) s += a.s2.tabT[((size_t)c * Q + q) * P2 + i] * vq[c * Q + q];
    yt[i] = s;
  }
  wave_sync();
  for (int i = lane; i < P2; i += 64) {
    double s = yt[i];
    if (a.s2.cor) {
      const int8_t *t = a.s2.cor + 3 * ((size_t)e * P2 + i);
      s = (double)t[1] * yt[i];
      if (i > 0) s += (double)t[-3 + 2] * yt[i - 1];       // T[i-1][i]
      if (i + 1 < P2) s += (double)t[3 + 0] * yt[i + 1];    // T[i+1][i]
    }
    a.ye[(size_t)e * P2 + i] = s;
  }
}

// Geometry data whose adj(J)^T / detJ is SDIM x DIM (2 + SDIM DIM rows): boundary (3, 2) and line (3, 1), (2, 1) elements with
// KIND 0 / 1 / 4 as above -- f_apply_hcurlhdiv_*, f_apply_hdivhcurl_*, f_apply_hcurl_* between two spaces, DIM components on
// both sides (MultAtBCx32 / 31 / 21: v = w detJ L^T C R u, L / R = the stored matrix or AdjJt32 / 31 / 21 of it) -- and, on
// every geometry, KIND 7: f_apply_hcurlh1d_* (hcurlh1d_33_qf.h:10-30, MultBAx*: v = w detJ C (adjJt u)), the DIM reference
// components of the trial gradient to the SDIM components of a vector H1 test space; each component uses the scalar value
// table of s2 and is one row [P2] of the element's E-vector block [SDIM][P2].
template <int KIND, int SDIM, int DIM>
__global__ __launch_bounds__(64 * kMixWaves) void mixed_embedded_kernel(const MixArgs a) {
  static_assert(KIND == 0 || KIND == 1 || KIND == 4 || KIND == 7, "apply forms only");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * kMixWaves + wave;
  if (e >= a.ne) return;  // no workgroup barriers below
  const int P1 = a.s1.P, P2 = a.s2.P, Q = a.Q;
  double *xa = smem + (size_t)wave * a.stride;
  double *xb = xa + P1;
  double *tmp = xb + P2;
  double *vq = tmp + max(P1, P2);
  constexpr int NV = KIND == 7 ? SDIM : DIM;

  mix_gather(a.s1, e, lane, a.x1, xa, tmp);
  const double *g = a.geom + ((size_t)(e / kEBm) * (2 + SDIM * DIM) * a.Qpad) * kEBm + (e % kEBm);
  for (int q = lane; q < Q; q += 64) {
    double u1[3];
    mix_eval(a.s1, Q, q, xa, u1);
    const int attr = a.c0.nattr > 0 ? max(1, (int)g[(size_t)q * kEBm]) : 1;
    const double wdetJ = g[((size_t)a.Qpad + q) * kEBm];
    double A[SDIM * DIM], C[SDIM * SDIM];
#pragma unroll
    for (int k = 0; k < SDIM * DIM; k++) A[k] = g[((size_t)(2 + k) * a.Qpad + q) * kEBm];
    {
      const double *m = a.c0.mat + SDIM * SDIM * coeff_index(a.c0, attr);  // CoeffUnpack2 / CoeffUnpack3, column-major
#pragma unroll
      for (int k = 0; k < SDIM * SDIM; k++) C[k] = m[k];
    }
    double T[SDIM * DIM];  // the contravariant map
    if (KIND == 0 || KIND == 1) {
      if (DIM == 2) {  // AdjJt32 (utils_32_qf.h:23-40)
        const double E = A[0] * A[0] + A[1] * A[1] + A[2] * A[2], G = A[3] * A[3] + A[4] * A[4] + A[5] * A[5];
        const double F = A[0] * A[3] + A[1] * A[4] + A[2] * A[5], d = sqrt(E * G - F * F);
#pragma unroll
        for (int k = 0; k < 3; k++) T[k] = (G * A[k] - F * A[3 + k]) / d, T[3 + k] = (E * A[3 + k] - F * A[k]) / d;
      } else {  // AdjJt21 / AdjJt31 (utils_21_qf.h:19-28, utils_31_qf.h:19-31)
        double n2 = 0.0;
#pragma unroll
        for (int i = 0; i < SDIM; i++) n2 += A[i] * A[i];
        const double d = sqrt(n2);
#pragma unroll
        for (int i = 0; i < SDIM; i++) T[i] = A[i] / d;
      }
    }
    const double *L = KIND == 0 ? T : A, *R = KIND == 1 ? T : A;
    double y[SDIM], z[SDIM];
#pragma unroll
    for (int i = 0; i < SDIM; i++) {
      y[i] = 0.0;
#pragma unroll
      for (int j = 0; j < DIM; j++) y[i] += R[i + SDIM * j] * u1[j];
    }
#pragma unroll
    for (int i = 0; i < SDIM; i++) {
      z[i] = 0.0;
#pragma unroll
      for (int j = 0; j < SDIM; j++) z[i] += C[i + SDIM * j] * y[j];
    }
    if (KIND == 7) {
#pragma unroll
      for (int i = 0; i < SDIM; i++) vq[i * Q + q] = wdetJ * z[i];
    } else {
#pragma unroll
      for (int j = 0; j < DIM; j++) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < SDIM; i++) v += L[i + SDIM * j] * z[i];
        vq[j * Q + q] = wdetJ * v;
      }
    }
  }
  wave_sync();
  double *yt = tmp;
  if (KIND == 7) {  // every component through the scalar table, one E-vector row each (no dof transformation on H1 spaces)
    for (int c = 0; c < NV; c++) {
      for (int i = lane; i < P2; i += 64) {
        double s = 0.0;
        for (int q = 0; q < Q; q++) s += a.s2.tabT[(size_t)q * P2 + i] * vq[c * Q + q];
        a.ye[((size_t)e * NV + c) * P2 + i] = s;
      }
    }
    return;
  }
  for (int i = lane; i < P2; i += 64) {
    double s = 0.0;
    for (int c = 0; c < DIM; c++)
      for (int q = 0; q < Q; q++) s += a.s2.tabT[((size_t)c * Q + q) * P2 + i] * vq[c * Q + q];
    yt[i] = s;
  }
  wave_sync();
  for (int i = lane; i < P2; i += 64) {
    double s = yt[i];
    if (a.s2.cor) {
      const int8_t *t = a.s2.cor + 3 * ((size_t)e * P2 + i);
      s = (double)t[1] * yt[i];
      if (i > 0) s += (double)t[-3 + 2] * yt[i - 1];       // T[i-1][i]
      if (i + 1 < P2) s += (double)t[3 + 0] * yt[i + 1];    // T[i+1][i]
    }
    a.ye[(size_t)e * P2 + i] = s;
  }
}

// H(curl) and H(div) sides enter with their value tables (Interp), an H1 side with its gradient table (Grad): the covariant
// map of H(curl) values is the map of gradients.
// `scalar`: the side enters a scalar QFunction (or is one component of a vector H1 space) with the values of a scalar space.
// `curl`: an H(curl) side enters with its curl table (Curl) instead of its values.
void build_side(const pa_restriction_desc &r, const pa_dense_basis_desc &b, int Q, int nc, MixedSide &sd, bool scalar,
                bool curl = false) {
  PA_REQUIRE(b.fe_type == PA_FE_HCURL || b.fe_type == PA_FE_HDIV || b.fe_type == PA_FE_H1, "unknown element type");
  PA_REQUIRE(!scalar || (nc == 1 && b.fe_type == PA_FE_H1), "scalar QFunctions take scalar (PA_FE_H1 descriptor) elements");
  const double *tab = ((b.fe_type == PA_FE_H1 && !scalar) || (curl && b.fe_type == PA_FE_HCURL)) ? b.deriv : b.interp;
  PA_REQUIRE(b.num_dofs > 0 && b.num_qpts == Q && tab, "basis does not match the quadrature rule, or has no value / gradient table");
  PA_REQUIRE(r.elem_size == b.num_dofs && r.offsets && r.lsize > 0, "restriction does not match the basis");
  PA_REQUIRE(!(r.orients && r.curl_orients), "restriction is either oriented or curl-oriented");
  const int P = b.num_dofs, ne = r.num_elem;
  sd.fe_type = b.fe_type, sd.P = P, sd.lsize = r.lsize, sd.nc = nc;
  std::vector<int32_t> sidx((size_t)ne * P);
  for (size_t k = 0; k < sidx.size(); k++) {
    const int32_t off = r.offsets[k];
    PA_REQUIRE(off >= 0 && off < r.lsize, "restriction offset out of range");
    sidx[k] = (r.orients && r.orients[k]) ? -1 - off : off;
  }
  sd.d_sidx = dev_upload(sidx.data(), sidx.size());
  sd.h_off.assign(r.offsets, r.offsets + (size_t)ne * P);
  if (r.curl_orients) sd.d_cor = dev_upload(r.curl_orients, (size_t)3 * ne * P);
  std::vector<double> F((size_t)nc * P * Q);
  for (int c = 0; c < nc; c++)
    for (int q = 0; q < Q; q++)
      for (int d = 0; d < P; d++) F[((size_t)c * P + d) * Q + q] = tab[((size_t)c * Q + q) * P + d];
  sd.d_tabF = dev_upload(F.data(), F.size());
  sd.d_tabT = dev_upload(tab, (size_t)nc * Q * P);
  // transpose map of the signed plain-layout index (counting sort by dof, element order preserved)
  std::vector<int32_t> tptr((size_t)r.lsize + 1, 0), tent((size_t)ne * P);
  for (size_t k = 0; k < (size_t)ne * P; k++) tptr[(size_t)r.offsets[k] + 1]++;
  for (int d = 0; d < r.lsize; d++) tptr[d + 1] += tptr[d];
  std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
  for (size_t k = 0; k < (size_t)ne * P; k++) {
    const bool flip = r.orients && r.orients[k];
    tent[fill[r.offsets[k]]++] = flip ? -1 - (int32_t)k : (int32_t)k;
  }
  sd.d_tptr = dev_upload(tptr.data(), tptr.size());
  sd.d_tent = dev_upload(tent.data(), tent.size());
}

void free_side(MixedSide &sd) {
  hipFree(sd.d_sidx), hipFree(sd.d_cor), hipFree(sd.d_tabF), hipFree(sd.d_tabT), hipFree(sd.d_tptr), hipFree(sd.d_tent);
}

MixSideDev dev_side(const MixedSide &sd) { return MixSideDev{sd.P, sd.nc, sd.d_sidx, sd.d_cor, sd.d_tabF, sd.d_tabT}; }

// transpose: A^T = B_1^T W^T B_2 -- the two sides change places, W^T is the other member of the QFunction pair (Jl^T C adj and
// adj^T C Jl are each other's transposes, adj^T C adj and the scalar mass their own) with the transposed coefficient, which
// CoeffHost::dev() hands out inside the TransposeScope of pa_op_mult_transpose
void launch(const MixedSub &ms, const double *x1, const double *x2, double *out, hipStream_t s, bool transpose = false) {
  MixArgs a;
  a.ne = ms.ne, a.Q = ms.Q, a.Qpad = ms.geom->Qpad, a.dim = ms.geom->dim;
  a.stride = (ms.s1.P + ms.s2.P + std::max(ms.s1.P, ms.s2.P) + 3 * ms.Q + 1) & ~1;
  a.geom = ms.geom->d_geom;
  a.s1 = dev_side(transpose ? ms.s2 : ms.s1), a.s2 = dev_side(transpose ? ms.s1 : ms.s2);
  a.c0 = ms.c0.dev(), a.c1 = ms.c1.dev();
  if (transpose && !ms.d_ye_t) ms.d_ye_t = dev_alloc<double>((size_t)ms.ne * ms.s1.P);
  a.x1 = x1, a.x2 = x2, a.ye = transpose ? ms.d_ye_t : ms.d_ye, a.out = out;
  const int kind = !transpose ? ms.kind : (ms.kind == 0 ? 1 : (ms.kind == 1 ? 0 : ms.kind));
  const size_t shm = sizeof(double) * (size_t)a.stride * kMixWaves;
  PA_REQUIRE(shm <= 64 * 1024, "element too large for the mixed-space kernels");
  const dim3 grid((ms.ne + kMixWaves - 1) / kMixWaves), block(64 * kMixWaves);
  const int dims = 10 * ms.geom->sdim + ms.geom->dim;
  if (kind == 7 || dims == 32 || dims == 31 || dims == 21) {
    PA_REQUIRE(kind != 7 || !transpose, "the gradient form has no transposed apply");
    bool done = true;
#define PA_MIX_EMB(K, SD, D)                                                                  \
  if (kind == K && dims == 10 * SD + D)                                                       \
    hipLaunchKernelGGL((mixed_embedded_kernel<K, SD, D>), grid, block, shm, s, a);            \
  else
    PA_MIX_EMB(0, 3, 2) PA_MIX_EMB(1, 3, 2) PA_MIX_EMB(4, 3, 2)
    PA_MIX_EMB(0, 3, 1) PA_MIX_EMB(1, 3, 1) PA_MIX_EMB(4, 3, 1)
    PA_MIX_EMB(0, 2, 1) PA_MIX_EMB(1, 2, 1) PA_MIX_EMB(4, 2, 1)
    PA_MIX_EMB(7, 3, 3) PA_MIX_EMB(7, 2, 2) PA_MIX_EMB(7, 3, 2) PA_MIX_EMB(7, 3, 1) PA_MIX_EMB(7, 2, 1)
    done = false;
#undef PA_MIX_EMB
    PA_REQUIRE(done, "QFunction not built for boundary / line elements");
    PA_HIP(hipGetLastError());
    return;
  }
  switch (kind) {
    case 0: hipLaunchKernelGGL(mixed_kernel<0>, grid, block, shm, s, a); break;
    case 1: hipLaunchKernelGGL(mixed_kernel<1>, grid, block, shm, s, a); break;
    case 2: hipLaunchKernelGGL(mixed_kernel<2>, grid, block, shm, s, a); break;
    case 3: hipLaunchKernelGGL(mixed_kernel<3>, grid, block, shm, s, a); break;
    case 4: hipLaunchKernelGGL(mixed_kernel<4>, grid, block, shm, s, a); break;
    case 5: hipLaunchKernelGGL(mixed_kernel<5>, grid, block, shm, s, a); break;
    case 6: hipLaunchKernelGGL(mixed_kernel<6>, grid, block, shm, s, a); break;
    case 8: hipLaunchKernelGGL(mixed_kernel<8>, grid, block, shm, s, a); break;
    default: throw Error("not a mixed-space QFunction");
  }
  PA_HIP(hipGetLastError());
}

}  // namespace

namespace {
// transpose map of a signed index list [n] into an L-vector of lsize entries (counting sort by dof, list order preserved)
void build_transpose_map(const std::vector<int32_t> &off, const uint8_t *orients, int lsize, MixedSide &sd) {
  std::vector<int32_t> tptr((size_t)lsize + 1, 0), tent(off.size());
  for (size_t k = 0; k < off.size(); k++) tptr[(size_t)off[k] + 1]++;
  for (int d = 0; d < lsize; d++) tptr[d + 1] += tptr[d];
  std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
  for (size_t k = 0; k < off.size(); k++) tent[fill[off[k]]++] = (orients && orients[k]) ? -1 - (int32_t)k : (int32_t)k;
  hipFree(sd.d_tptr), hipFree(sd.d_tent);
  sd.d_tptr = dev_upload(tptr.data(), tptr.size());
  sd.d_tent = dev_upload(tent.data(), tent.size());
}
}  // namespace

MixedSub *make_mixed_sub(pa_geom *geom, const pa_restriction_desc &r1, const pa_dense_basis_desc &b1,
                         const pa_restriction_desc &r2, const pa_dense_basis_desc &b2, int qf, const void *ctx,
                         size_t ctx_size) {
  PA_REQUIRE(geom && geom->eb == kEBm && geom->dim >= 1 && geom->dim <= geom->sdim && geom->sdim <= 3,
             "mixed-space operators need geometry data from pa_geom_create_dense");
  PA_REQUIRE(r1.num_elem == geom->ne && r2.num_elem == geom->ne, "restrictions do not match the mesh");
  const int dim = geom->dim, sdim = geom->sdim, dims = 10 * sdim + dim;
  int kind = -1, qdims = 0;  // the QFunction family and the member (10 space_dim + dim; 0: any geometry)
  switch (qf) {
    case PA_QF_HCURLHDIV_33: kind = 0, qdims = 33; break;
    case PA_QF_HCURLHDIV_22: kind = 0, qdims = 22; break;
    case PA_QF_HCURLHDIV_32: kind = 0, qdims = 32; break;
    case PA_QF_HCURLHDIV_31: kind = 0, qdims = 31; break;
    case PA_QF_HCURLHDIV_21: kind = 0, qdims = 21; break;
    case PA_QF_HDIVHCURL_33: kind = 1, qdims = 33; break;
    case PA_QF_HDIVHCURL_22: kind = 1, qdims = 22; break;
    case PA_QF_HDIVHCURL_32: kind = 1, qdims = 32; break;
    case PA_QF_HDIVHCURL_31: kind = 1, qdims = 31; break;
    case PA_QF_HDIVHCURL_21: kind = 1, qdims = 21; break;
    case PA_QF_HCURLHDIV_ERROR_33: kind = 2, qdims = 33; break;
    case PA_QF_HCURLHDIV_ERROR_22: kind = 2, qdims = 22; break;
    case PA_QF_HDIVHCURL_ERROR_33: kind = 3, qdims = 33; break;
    case PA_QF_HDIVHCURL_ERROR_22: kind = 3, qdims = 22; break;
    case PA_QF_HCURL_33: kind = 4, qdims = 33; break;
    case PA_QF_HCURL_22: kind = 4, qdims = 22; break;
    case PA_QF_HCURL_32: kind = 4, qdims = 32; break;
    case PA_QF_HCURL_31: kind = 4, qdims = 31; break;
    case PA_QF_HCURL_21: kind = 4, qdims = 21; break;
    case PA_QF_H1_1: kind = 5; break;
    case PA_QF_L2H1_ERROR: kind = 6; break;
    case PA_QF_HDIV_33: kind = 8, qdims = 33; break;
    default: throw Error("not a mixed-space QFunction");
  }
  const bool scalar = kind == 5 || kind == 6;  // dimension-independent: reads w detJ only
  PA_REQUIRE(scalar ? dim == sdim : qdims == dims, "QFunction does not match the dimension of the geometry data");
  const bool err = kind == 2 || kind == 3 || kind == 6;
  // first space / second space by the Piola map the QFunction applies to each input: covariant (H(curl) values, H1
  // gradients) or contravariant (H(div) values)
  auto covariant = [](const pa_dense_basis_desc &b) { return b.fe_type == PA_FE_HCURL || b.fe_type == PA_FE_H1; };
  const bool cov1 = kind == 0 || kind == 2 || kind == 4, cov2 = kind == 1 || kind == 3 || kind == 4;
  // (f_apply_hdiv_33: the curls of an H(curl) space are contravariant like the values of an H(div) one)
  auto contra8 = [](const pa_dense_basis_desc &b) { return b.fe_type == PA_FE_HDIV || (b.fe_type == PA_FE_HCURL && b.deriv); };
  PA_REQUIRE(scalar || (kind == 8 ? (contra8(b1) && contra8(b2))
                                  : ((cov1 ? covariant(b1) : b1.fe_type == PA_FE_HDIV) && (cov2 ? covariant(b2) : b2.fe_type == PA_FE_HDIV))),
             "element types do not match the QFunction (vecfemass.cpp:88-101, mixedvecgrad.cpp:43-76, mixedveccurl.cpp:41-58)");
  PA_REQUIRE(ctx && ctx_size >= 16 && ctx_size % 8 == 0, "bad coefficient context");
  auto *ms = new MixedSub;
  try {
    ms->geom = geom;
    geom->refcount++;
    ms->ne = geom->ne, ms->Q = geom->Q, ms->qf = qf, ms->kind = kind, ms->error = err;
    build_side(r1, b1, geom->Q, scalar ? 1 : dim, ms->s1, scalar, kind == 8);
    build_side(r2, b2, geom->Q, scalar ? 1 : dim, ms->s2, scalar, kind == 8);
    parse_coeff(ctx, ctx_size, scalar ? 1 : sdim, ms->c0, 0);
    if (err) parse_coeff(ctx, ctx_size, scalar ? 1 : sdim, ms->c1, ms->c0.slots);  // PopulateCoefficientContext(dim, first, dim, second)
    if (!err) ms->d_ye = dev_alloc<double>((size_t)ms->ne * ms->s2.P);
  } catch (...) {
    free_mixed_sub(ms);
    throw;
  }
  return ms;
}

// GradientIntegrator (fem/integ/grad.cpp:16-72): trial = a scalar H1 space (Grad), test = a vector H1 space with space_dim
// components (Interp); `r2` / `b2` describe ONE component of the test space (its scalar value table; offsets = the L-vector
// index of component 0), component c of a dof lives comp_stride entries further (restriction.cpp:137-142: 1 for byVDIM, where
// the offsets are already multiplied by the vector dimension, the number of dofs for byNODES); r2.lsize is the size of the whole
// vector L-vector.
MixedSub *make_mixed_gradient_sub(pa_geom *geom, const pa_restriction_desc &r1, const pa_dense_basis_desc &b1,
                                  const pa_restriction_desc &r2, const pa_dense_basis_desc &b2, int comp_stride, int qf,
                                  const void *ctx, size_t ctx_size) {
  PA_REQUIRE(geom && geom->eb == kEBm && geom->dim >= 1 && geom->dim <= geom->sdim && geom->sdim <= 3,
             "the gradient form needs geometry data from pa_geom_create_dense");
  PA_REQUIRE(r1.num_elem == geom->ne && r2.num_elem == geom->ne, "restrictions do not match the mesh");
  const int dim = geom->dim, sdim = geom->sdim, dims = 10 * sdim + dim;
  const int want = dims == 33   ? PA_QF_HCURLH1D_33
                   : dims == 22 ? PA_QF_HCURLH1D_22
                   : dims == 32 ? PA_QF_HCURLH1D_32
                   : dims == 31 ? PA_QF_HCURLH1D_31
                                : PA_QF_HCURLH1D_21;
  PA_REQUIRE(qf == want, "QFunction does not match the dimension of the geometry data");
  PA_REQUIRE(b1.fe_type == PA_FE_H1 && b2.fe_type == PA_FE_H1 && b1.deriv && b2.interp,
             "GradientIntegrator requires a scalar H1 trial space (gradient table) and an H1 test space (value table)!");
  PA_REQUIRE(comp_stride >= 1 && !r2.orients && !r2.curl_orients, "bad component stride / oriented H1 restriction");
  PA_REQUIRE(ctx && ctx_size >= 16 && ctx_size % 8 == 0, "bad coefficient context");
  auto *ms = new MixedSub;
  try {
    ms->geom = geom;
    geom->refcount++;
    ms->ne = geom->ne, ms->Q = geom->Q, ms->qf = qf, ms->kind = 7, ms->error = false;
    build_side(r1, b1, geom->Q, dim, ms->s1, false);
    build_side(r2, b2, geom->Q, 1, ms->s2, true);
    // the E-vector block of an element is [sdim][P2]: its transpose map over the whole vector L-vector
    const int P2 = ms->s2.P;
    std::vector<int32_t> off((size_t)ms->ne * sdim * P2);
    for (int e = 0; e < ms->ne; e++)
      for (int c = 0; c < sdim; c++)
        for (int i = 0; i < P2; i++) {
          const int64_t v = (int64_t)r2.offsets[(size_t)e * P2 + i] + (int64_t)c * comp_stride;
          PA_REQUIRE(v < r2.lsize, "component offset out of range (comp_stride and lsize of the vector space)");
          off[((size_t)e * sdim + c) * P2 + i] = (int32_t)v;
        }
    build_transpose_map(off, nullptr, r2.lsize, ms->s2);
    ms->s2.h_off = off;
    parse_coeff(ctx, ctx_size, sdim, ms->c0, 0);
    ms->d_ye = dev_alloc<double>((size_t)ms->ne * sdim * P2);
  } catch (...) {
    free_mixed_sub(ms);
    throw;
  }
  return ms;
}

void free_mixed_sub(MixedSub *ms) {
  if (!ms) return;
  free_side(ms->s1), free_side(ms->s2);
  hipFree(ms->d_ye), hipFree(ms->d_ye_t);
  hipFree(ms->c0.d_attr_mat), hipFree(ms->c0.d_mat), hipFree(ms->c0.d_mat_t);
  hipFree(ms->c1.d_attr_mat), hipFree(ms->c1.d_mat), hipFree(ms->c1.d_mat_t);
  pa_geom_destroy(static_cast<pa_geom *>(ms->geom));
  delete ms;
}

void launch_mixed_apply(const MixedSub &ms, const double *x, double *y, bool accumulate, hipStream_t s, bool transpose) {
  PA_REQUIRE(!ms.error, "error integrators have no apply");
  launch(ms, x, nullptr, nullptr, s, transpose);
  const MixedSide &out = transpose ? ms.s1 : ms.s2;
  launch_et_gather_raw(out.lsize, out.d_tptr, out.d_tent, transpose ? ms.d_ye_t : ms.d_ye, y, accumulate, s);
}

void launch_mixed_error(const MixedSub &ms, const double *u1, const double *u2, double *out, hipStream_t s) {
  PA_REQUIRE(ms.error, "not an error integrator");
  launch(ms, u1, u2, out, s);
}

}  // namespace pa
