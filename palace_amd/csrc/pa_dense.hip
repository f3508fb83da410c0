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
#include <atomic>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "pa_device.hpp"
#include "pa_internal.hpp"

#ifndef PA_DENSE_XEARLY
#define PA_DENSE_XEARLY 0
#endif

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
  MODE_MASS = 5,      // H1 mass                   f_apply_h1_1
  // 2-D (fast path only: tables resident in LDS + packed D)
  MODE_CURL2 = 6,     // 2-D ND curl-curl          f_apply_l2_1 on the scalar curl (q_w input)
  MODE_VMASS2 = 7,    // 2-D ND mass               f_apply_hcurl_22
  MODE_CURLMASS2 = 8, // 2-D ND curl-curl + mass   f_apply_hdivmass_22 / _32
  MODE_DIFF2 = 9,     // 2-D H1 diffusion          f_apply_hcurl_22 / _32 on grad u
  MODE_DIFFMASS2 = 10, // 2-D H1 diffusion + mass   f_apply_hcurlmass_22 / _32
  // (MODE_MASS serves the 2-D and 1-D H1 mass too: f_apply_h1_1 only reads w detJ)
  // line elements (boundaries of plane problems, curves in space): every D is a scalar per point
  MODE_VMASS1 = 11,    // ND mass on a line         f_apply_hcurl_21 / _31
  MODE_DIFF1 = 12,     // H1 diffusion on a line    f_apply_hcurl_21 / _31 on du/dxi
  MODE_DIFFMASS1 = 13  // H1 diffusion + mass       f_apply_hcurlmass_21 / _31
};

template <int MODE>
struct ModeTraits {
  static constexpr int NCI = (MODE == MODE_VMASS || MODE == MODE_CURLMASS) ? 3
                             : (MODE == MODE_VMASS2 || MODE == MODE_CURLMASS2) ? 2
                             : (MODE == MODE_DIFFMASS || MODE == MODE_MASS || MODE == MODE_DIFFMASS2 || MODE == MODE_VMASS1 ||
                                MODE == MODE_DIFFMASS1) ? 1 : 0;
  static constexpr int NCD = (MODE == MODE_VMASS || MODE == MODE_MASS || MODE == MODE_VMASS2 || MODE == MODE_VMASS1) ? 0
                             : (MODE == MODE_CURL2 || MODE == MODE_CURLMASS2 || MODE == MODE_DIFF1 || MODE == MODE_DIFFMASS1) ? 1
                             : (MODE == MODE_DIFF2 || MODE == MODE_DIFFMASS2) ? 2 : 3;
  static constexpr int NCT = NCI + NCD;
};

int mode_of(int fe_type, int qf, int dim, int sdim) {
  if (dim == 1) {  // line elements in the plane (21) or in space (31)
    const int q_vec = sdim == 3 ? PA_QF_HCURL_31 : PA_QF_HCURL_21, q_pair = sdim == 3 ? PA_QF_HCURLMASS_31 : PA_QF_HCURLMASS_21;
    if (fe_type == PA_FE_HCURL && qf == q_vec) return MODE_VMASS1;
    if (fe_type == PA_FE_H1 && qf == PA_QF_H1_1) return MODE_MASS;
    if (fe_type == PA_FE_H1 && qf == q_vec) return MODE_DIFF1;
    if (fe_type == PA_FE_H1 && qf == q_pair) return MODE_DIFFMASS1;
    throw Error("QFunction does not match a line element");
  }
  if (dim == 2) {  // the same applies for plane elements and for boundary elements in 3-D: only D differs (3x2 geometry)
    const bool bdr = sdim == 3;
    if (fe_type == PA_FE_HCURL) {
      if (qf == PA_QF_L2_1) return MODE_CURL2;
      if (qf == (bdr ? PA_QF_HCURL_32 : PA_QF_HCURL_22)) return MODE_VMASS2;
      if (qf == (bdr ? PA_QF_HDIVMASS_32 : PA_QF_HDIVMASS_22)) return MODE_CURLMASS2;
      throw Error(bdr ? "QFunction does not match an H(curl) boundary element" : "QFunction does not match a 2-D H(curl) element");
    }
    if (qf == PA_QF_H1_1) return MODE_MASS;
    if (qf == (bdr ? PA_QF_HCURL_32 : PA_QF_HCURL_22)) return MODE_DIFF2;
    if (qf == (bdr ? PA_QF_HCURLMASS_32 : PA_QF_HCURLMASS_22)) return MODE_DIFFMASS2;
    throw Error(bdr ? "QFunction does not match an H1 boundary element" : "QFunction does not match a 2-D H1 element");
  }
  if (fe_type == PA_FE_HCURL) {
    if (qf == PA_QF_L2_1) return MODE_CURL2;  // (reached through the H(div) alias: div-div, a scalar derivative in 3-D)
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
  nci = (mode == MODE_VMASS || mode == MODE_CURLMASS) ? 3
        : (mode == MODE_VMASS2 || mode == MODE_CURLMASS2) ? 2
        : (mode == MODE_DIFFMASS || mode == MODE_MASS || mode == MODE_DIFFMASS2 || mode == MODE_VMASS1 || mode == MODE_DIFFMASS1) ? 1
        : 0;
  ncd = (mode == MODE_VMASS || mode == MODE_MASS || mode == MODE_VMASS2 || mode == MODE_VMASS1) ? 0
        : (mode == MODE_CURL2 || mode == MODE_CURLMASS2 || mode == MODE_DIFF1 || mode == MODE_DIFFMASS1) ? 1
        : (mode == MODE_DIFF2 || mode == MODE_DIFFMASS2) ? 2 : 3;
}

// position of (dof slot s, lane) of a block's E-vector: lane = kq * 16 + element, dof = 4 s + kq
__device__ __forceinline__ int ye_pos(const int rows, const int KP, const int s, const int lane) {
  return rows ? (lane & 15) * (4 * KP) + 4 * s + (lane >> 4) : s * 64 + lane;
}
inline size_t ye_pos_host(bool rows, int KP, int e, int d) {  // (element e, local dof d) in the whole E-vector
  const size_t blk = (size_t)(e / kEB) * 4 * KP * kEB;
  return rows ? blk + (size_t)(e % kEB) * 4 * KP + d : blk + (size_t)d * kEB + (e % kEB);
}

struct DenseArgs {
  int ne, nb, P, Q, Qpad, nch, KP;
  const int32_t *idx;
  const uint16_t *co;  // 2-bit fields {sub, main, super} of row d of T_e and {T[d-1][d], T[d+1][d]} of column d
  const uint32_t *co2; // the same, the words of dof slots 2 s and 2 s + 1 of a lane in one register: [nb][KP / 2][64] (resident kernel)
  const double *geom;
  const double *qw;     // quadrature weights (2-D curl-curl)
  int contra;           // vector mass with the contravariant map of H(div) values (f_apply_hdiv_22 | _32 | _21 | _31: AdjJt of the
                        // stored adj(J)^T / detJ) instead of adj(J)^T / detJ; with a pair mode: the div-div + mass forms f_apply_l2mass_*
  const double *Tf, *Tt;
  const double *L;      // resident form of the tables: [rows][S], rows in tile order (see make_dense_sub)
  const double *qdata;  // packed pre-assembled D: [nb][ncq][Qpad][16]
  int ncq, Q4;  // q-data components and its point stride (Q rounded up to a multiple of 4)
  const uint8_t *affine;  // non-null: every element has a constant Jacobian, D_q = (w_q / w_0) D_0
  const int32_t *blist;   // optional list of the element blocks this launch works on (nblist entries): a mesh with affine and
  int nblist;             // curved blocks runs the affine kernel on the former and the general one on the latter
  // complex form (CPLX): imaginary parts of x and of the E-vector; packed D of the imaginary operator with the offsets of its
  // mass / curl-curl components (-1: no such term)
  const double *x1;
  double *ye1;
  const double *qdata_i;
  int ncq_i, qi_mass, qi_curl;
  const double *wrel;     // [Q4] w_q / w_0
  int dbg;  // ablation bits (PA_ABLATION builds only)
  const double *x;
  double *ye;
  int ye_rows;  // E-vector layout inside a block of 16 elements: 0 = [dof][element] (one 128-byte row per dof: what the kernels' lanes
                // hold side by side), 1 = [element][dof] (round 6: the gather reads the dofs of an edge / a face of one element from
                // ONE 64-byte sector instead of one sector per 8-byte entry -- it is bound by those sectors)
  // split vectors (multi-rank applies without L-vector copies): local dofs >= nsplit are read from xg0 / xg1 (the parity of
  // *xg_sel picks the buffer; both stored shifted by -nsplit); nsplit = INT_MAX otherwise
  int nsplit;
  const double *xg0, *xg1;
  const unsigned long long *xg_sel;
  CoeffDev c0, c1;
};
// where dof d of the input is read (d without its flag bits)
__device__ __forceinline__ const double *dense_xbase(const DenseArgs &a, const double *x, const double *xg, const int d) {
  return d < a.nsplit ? x : xg;
}
__device__ __forceinline__ const double *dense_ghosts(const DenseArgs &a) {
  if (a.nsplit == 0x7fffffff) return a.x;
  return ((a.xg_sel ? *a.xg_sel : 0ull) & 1ull) ? a.xg1 : a.xg0;
}

// 2-bit two's-complement field k of a packed curl-orientation word: -1, 0 or 1
__device__ __forceinline__ double co_field(const int c, const int k) {
  return (double)((c << (30 - 2 * k)) >> 30);
}

// the same from a register holding two 16-bit words (word `odd` = 0 | 1): the left shift drops the other word's bits
__device__ __forceinline__ double co_field_pk(const unsigned w, const int odd, const int k) {
  return (double)(((int)(w << (30 - 2 * k - 16 * odd))) >> 30);
}

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
constexpr int kDenseThreads = 64 * kDenseWaves;

// Components of field F of a mode (field 0 = the interp part if the mode has one, then the
// curl / gradient part) and the number of components before it.
template <int MODE, int F>
struct FieldTraits {
  using M = ModeTraits<MODE>;
  static constexpr int NF = (M::NCI > 0 ? 1 : 0) + (M::NCD > 0 ? 1 : 0);
  static constexpr int NC = (NF == 2) ? (F == 0 ? M::NCI : M::NCD) : M::NCT;
  static constexpr int CB = (NF == 2 && F == 1) ? M::NCI : 0;
};

// D on the components of one field at one quadrature point (in place).  The paired QFunctions
// (hdivmass_33, hcurlmass_33) act on their two inputs independently, so the fields can be processed
// one after the other with the same geometry registers.
template <int MODE, int F>
__device__ __forceinline__ void dense_D_field(const DenseArgs &a, const double wdetJ, const double (&adj)[9],
                                              const int attr, double *v) {
  double Cm[9];
  constexpr bool second = (F == 1);
  if (MODE == MODE_CURL || (MODE == MODE_CURLMASS && second)) {  // hdiv_33_qf.h:10-30, hdivmass_33_qf.h:30-42
    double Jl[9];
    coeff_unpack3(second ? a.c1 : a.c0, attr, Cm);
    adjJt33(adj, Jl);
    mult_AtBCx33(Jl, Cm, Jl, v[0], v[1], v[2], wdetJ, v[0], v[1], v[2]);
  } else if (MODE == MODE_VMASS || MODE == MODE_DIFF || MODE == MODE_CURLMASS || (MODE == MODE_DIFFMASS && second)) {
    coeff_unpack3(second ? a.c1 : a.c0, attr, Cm);  // hcurl_33_qf.h:10-28
    mult_AtBCx33(adj, Cm, adj, v[0], v[1], v[2], wdetJ, v[0], v[1], v[2]);
  } else {  // scalar mass: h1_1_qf.h, first half of hcurlmass_33_qf.h
    v[0] *= a.c0.mat[coeff_index(a.c0, attr)] * wdetJ;
  }
}

// One field of one 16-point chunk: stage the forward fragments in LDS, B, D, stage the transposed
// fragments, B^T.  `stage` arrives holding this field's forward fragments (prefetched) and leaves
// holding the next field's (if any).
template <int PT, int MODE, int F>
__device__ __forceinline__ void dense_field(const DenseArgs &a, const int lane, const int tid, const int,
                                            double *__restrict__ tab, const double *__restrict__ tf_next,
                                            const bool has_next, const int n_next, const double *__restrict__ tt,
                                            const double (&u)[4 * PT], const double (&gd)[4][10],
                                            const int (&attr)[4], double4_t (&yacc)[PT],
                                            double (&stage)[3 * PT], const int n_this) {
  using FT = FieldTraits<MODE, F>;
  constexpr int NC = FT::NC, KPMAX = 4 * PT, NST = 3 * PT, KP = KPMAX;
  __syncthreads();  // every wave is done reading the previous fragments
#pragma unroll
  for (int r = 0; r < NST; r++) {
    const int i = tid + kDenseThreads * r;
    if (i < n_this) tab[i] = stage[r];
  }
  __syncthreads();
  // ---- B: NC tiles of 16 rows = 4 points x (4 NC) (point group, component) pairs
  double4_t acc[NC];
#pragma unroll
  for (int t = 0; t < NC; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KPMAX; s++) {
    if (s < KP) {
#pragma unroll
      for (int t = 0; t < NC; t++)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(tab[(s * NC + t) * 64 + lane], u[s], acc[t], 0, 0, 0);
    }
  }
  // transposed fragments on their way while D runs
  constexpr int n_t = 4 * NC * PT * 64;
#pragma unroll
  for (int r = 0; r < NST; r++) {
    const int i = tid + kDenseThreads * r;
    if (i < n_t) stage[r] = tt[i];
  }
  // ---- D
#pragma unroll
  for (int gl = 0; gl < 4; gl++) {
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = acc[(gl * NC + k) >> 2][(gl * NC + k) & 3];
    double adj[9];
#pragma unroll
    for (int k = 0; k < 9; k++) adj[k] = gd[gl][1 + k];
    dense_D_field<MODE, F>(a, gd[gl][0], adj, attr[gl], v);
#pragma unroll
    for (int k = 0; k < NC; k++) acc[(gl * NC + k) >> 2][(gl * NC + k) & 3] = v[k];
  }
  __syncthreads();  // forward fragments no longer needed
#pragma unroll
  for (int r = 0; r < NST; r++) {
    const int i = tid + kDenseThreads * r;
    if (i < n_t) tab[i] = stage[r];
  }
  __syncthreads();
  // next field's forward fragments on their way while B^T runs
  if (has_next) {
#pragma unroll
    for (int r = 0; r < NST; r++) {
      const int i = tid + kDenseThreads * r;
      if (i < n_next) stage[r] = tf_next[i];
    }
  }
  // ---- B^T: the C layout above is the B-operand layout of the transposed product
#pragma unroll
  for (int pi = 0; pi < 4 * NC; pi++) {
#pragma unroll
    for (int pt = 0; pt < PT; pt++)
      yacc[pt] = __builtin_amdgcn_mfma_f64_16x16x4f64(tab[(pi * PT + pt) * 64 + lane], acc[pi >> 2][pi & 3], yacc[pt], 0, 0, 0);
  }
}

template <int PT, int MODE>
__global__ __launch_bounds__(kDenseThreads, (PT <= 4 ? 2 : 1)) void dense_apply_kernel(const DenseArgs a) {
  using M = ModeTraits<MODE>;
  using F0 = FieldTraits<MODE, 0>;
  using F1 = FieldTraits<MODE, 1>;
  constexpr int NCT = M::NCT, KPMAX = 4 * PT, NF = F0::NF;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b_raw = blockIdx.x * kDenseWaves + wave;
  const bool valid = b_raw < a.nb;  // surplus waves of the last workgroup redo the last block (barriers)
  const int b = valid ? b_raw : a.nb - 1;
  const int j = lane & 15, kq = lane >> 4;
  constexpr int KP = 4 * PT;  // the host pads every element block to 4 PT dof slots

  // forward fragments of the first field, requested before anything else
  double stage[3 * PT];
  const int n_f0 = KP * F0::NC * 64, n_f1 = KP * F1::NC * 64;
#pragma unroll
  for (int r = 0; r < 3 * PT; r++) {
    const int i = tid + kDenseThreads * r;
    if (i < n_f0) stage[r] = a.Tf[i];
  }

  // ---- E: gather straight into the MFMA B-operand layout
  double u[KPMAX];
  const double *xgh = dense_ghosts(a);
  const int32_t *idx = a.idx + (size_t)b * KP * 64;
#pragma unroll
  for (int s = 0; s < KPMAX; s++) {
    u[s] = 0.0;
    if (s < KP) {
      const int sg = idx[s * 64 + lane];
      const int d = sg >= 0 ? sg : -1 - sg;
      const double xv = (d & kEssBit) ? 0.0 : dense_xbase(a, a.x, xgh, d & ~kEssBit)[d & ~kEssBit];
      u[s] = sg >= 0 ? xv : -xv;
    }
  }
  const uint16_t *co = a.co ? a.co + (size_t)b * KP * 64 : nullptr;
  double *sm = smem + (size_t)wave * KPMAX * 64;
  double *tab = smem + (a.co ? (size_t)kDenseWaves * KPMAX * 64 : 0);
  if (co) {  // curl-oriented: u_e = T x_e, T tridiagonal (restriction.cpp:299-369)
#pragma unroll
    for (int s = 0; s < KPMAX; s++)
      if (s < KP) sm[s * 64 + lane] = u[s];
    wave_sync();
#pragma unroll
    for (int s = 0; s < KPMAX; s++) {
      if (s < KP) {
        const int c = co[s * 64 + lane];
        const int dof = 4 * s + kq;
        // out-of-range neighbours have a zero coefficient: clamp the address instead of branching
        const double lo = sm[max(dof - 1, 0) * 16 + j];
        const double hi = sm[min(dof + 1, 4 * KP - 1) * 16 + j];
        u[s] = co_field(c, 0) * lo + co_field(c, 1) * u[s] + co_field(c, 2) * hi;
      }
    }
    wave_sync();
  }

  double4_t yacc[PT];
#pragma unroll
  for (int pt = 0; pt < PT; pt++) yacc[pt] = double4_t{0.0, 0.0, 0.0, 0.0};

  const size_t tf_chunk = (size_t)KP * NCT * 64, tt_chunk = (size_t)4 * NCT * PT * 64;
  for (int c = 0; c < a.nch; c++) {
    // geometry of this lane's 4 points of the chunk (q = 16 c + 4 gl + kq)
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
    const double *tf = a.Tf + c * tf_chunk, *tt = a.Tt + c * tt_chunk;
    const bool more = c + 1 < a.nch;
    if (NF == 1) {
      dense_field<PT, MODE, 0>(a, lane, tid, KP, tab, tf + tf_chunk, more, n_f0, tt, u, gd, attr, yacc, stage, n_f0);
    } else {
      dense_field<PT, MODE, 0>(a, lane, tid, KP, tab, tf + n_f0, true, n_f1, tt, u, gd, attr, yacc, stage, n_f0);
      dense_field<PT, MODE, 1>(a, lane, tid, KP, tab, tf + tf_chunk, more, n_f0, tt + (size_t)4 * F0::NC * PT * 64, u, gd,
                               attr, yacc, stage, n_f1);
    }
  }
  if (!valid) return;

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
        const int cm = co[s * 64 + lane];
        ye[ye_pos(a.ye_rows, KP, s, lane)] = co_field(cm, 1) * yacc[s >> 2][s & 3] + co_field(cm, 3) * sm[max(dof - 1, 0) * 16 + j] +
                            co_field(cm, 4) * sm[min(dof + 1, 4 * KP - 1) * 16 + j];
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KPMAX; s++)
      if (s < KP) ye[ye_pos(a.ye_rows, KP, s, lane)] = yacc[s >> 2][s & 3];
  }
}

// ---- fast path: tables resident in LDS, pre-assembled packed D, persistent workgroups ---------------
// One copy of the tables, L[row][S] with rows in tile order and the row stride S = 18 (mod 32) doubles,
// serves both products: the forward A operand reads L[16 t + i][4 s + kq] (bank = 18 i + kq: all 32
// lanes of a half-wave distinct) and the transposed one L[4 pi + kq][16 pt + i] (two banks shared by
// two lanes).  No barrier after the initial load: every wave walks its own element blocks.
template <int PT>
struct ResidentStride {
  static constexpr int S = 16 * PT + ((PT & 1) ? 2 : 18);
};

constexpr int kResWaves = 8;

template <int MODE, int F>
__device__ __forceinline__ void dense_D_packed(const double *m, double *v) {
  using FT = FieldTraits<MODE, F>;
  if (FT::NC == 3) {
    sym_mv(m, v[0], v[1], v[2], v[0], v[1], v[2]);
  } else if (FT::NC == 2) {  // packed symmetric 2x2 {00, 01, 11}
    const double x0 = v[0], x1 = v[1];
    v[0] = m[0] * x0 + m[1] * x1;
    v[1] = m[1] * x0 + m[2] * x1;
  } else {
    v[0] *= m[0];
  }
}

// wrel != nullptr (affine elements): qd[0] holds the D of point 0, the D of point group gl is wrel[4 gl] times that
template <int PT, int MODE, int F>
__device__ __forceinline__ void resident_field(const double *__restrict__ Lf, const double *__restrict__ Lb, const int,
                                               const double (&u)[4 * PT], const double (*qd)[6],
                                               double4_t (&yacc)[PT], const double *__restrict__ wrel = nullptr,
                                               const double *qdi = nullptr, const double sg = 0.0,
                                               const double (*qdi4)[6] = nullptr) {
  using FT = FieldTraits<MODE, F>;
  constexpr int NC = FT::NC, KPMAX = 4 * PT, S = ResidentStride<PT>::S, KP = KPMAX;
  double4_t acc[NC];
#pragma unroll
  for (int t = 0; t < NC; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KPMAX; s++) {
    if (s < KP) {
#pragma unroll
      for (int t = 0; t < NC; t++)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lf[16 * t * S + 4 * s], u[s], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int gl = 0; gl < 4; gl++) {
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = acc[(gl * NC + k) >> 2][(gl * NC + k) & 3];
    if (wrel) {
      if (qdi) {  // complex form: (D_r + i D_i)(v_r + i v_i), the other part of v from the lane of the same element 8 columns away
        double vp[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) vp[k] = __shfl_xor(v[k], 8, 64);
        dense_D_packed<MODE, F>(qd[0], v);
        dense_D_packed<MODE, F>(qdi, vp);
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] += sg * vp[k];
      } else {
        dense_D_packed<MODE, F>(qd[0], v);
      }
      const double w = wrel[4 * gl];
#pragma unroll
      for (int k = 0; k < NC; k++) v[k] *= w;
    } else if (qdi4) {  // complex form on curved elements: the D of both operators at every point group
      double vp[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) vp[k] = __shfl_xor(v[k], 8, 64);
      dense_D_packed<MODE, F>(qd[gl], v);
      dense_D_packed<MODE, F>(qdi4[gl], vp);
#pragma unroll
      for (int k = 0; k < NC; k++) v[k] += sg * vp[k];
    } else {
      dense_D_packed<MODE, F>(qd[gl], v);
    }
#pragma unroll
    for (int k = 0; k < NC; k++) acc[(gl * NC + k) >> 2][(gl * NC + k) & 3] = v[k];
  }
#pragma unroll
  for (int pi = 0; pi < 4 * NC; pi++) {
#pragma unroll
    for (int pt = 0; pt < PT; pt++)
      yacc[pt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lb[4 * pi * S + 16 * pt], acc[pi >> 2][pi & 3], yacc[pt], 0, 0, 0);
  }
}

// q-data of the 4 point groups of one chunk; `ng` = valid groups left (the last chunk may be partial)
template <int MODE, int F>
__device__ __forceinline__ void load_qd(const double *__restrict__ q, const size_t cs, const int ng, double (&qd)[4][6]) {
  constexpr int NQ = FieldTraits<MODE, F>::NC == 3 ? 6 : (FieldTraits<MODE, F>::NC == 2 ? 3 : 1);
#pragma unroll
  for (int gl = 0; gl < 4; gl++)
#pragma unroll
    for (int k = 0; k < NQ; k++) qd[gl][k] = (gl < ng) ? q[k * cs + gl * 64] : 0.0;
}

// six rows (one symmetric 3 x 3 D) of the imaginary operator's q-data for the 4 point groups of a chunk; row0 < 0: that operator
// has no such term
__device__ __forceinline__ void load_qd6(const double *__restrict__ q, const int row0, const size_t cs, const int ng,
                                         double (&qd)[4][6]) {
#pragma unroll
  for (int gl = 0; gl < 4; gl++)
#pragma unroll
    for (int k = 0; k < 6; k++) qd[gl][k] = (row0 >= 0 && gl < ng) ? q[(size_t)(row0 + k) * cs + gl * 64] : 0.0;
}

// AFFINE: every block consists of elements with a constant Jacobian (dense_affine_kernel): the D of a point is the D of point 0
// times the relative quadrature weight -- 6 values per field and element instead of 6 Q, and no q-data registers to rotate.
#ifndef PA_DENSE_AFF_WAVES
#define PA_DENSE_AFF_WAVES 12
#endif
constexpr int kAffWaves = PA_DENSE_AFF_WAVES;  // the affine form needs fewer registers: three waves per SIMD
// CPLX (curl-curl + mass blocks, all affine or all curved): y = (A_r + i A_i) x for two operators on the same space and geometry.  A work unit is half
// an element block: the 16 element columns of the matrix-core products carry 8 elements x {real, imaginary} part of x; both
// parts read the element's index words and the D of both operators (one HBM read), the tables in LDS serve both as before,
// and the D stage combines the two parts across the columns of an element.  One pass instead of four.  (Curved blocks: the D of
// the imaginary operator is read chunk by chunk right before its use -- no registers left to prefetch it.)
// LIST: the launch works on the element blocks listed in a.blist (meshes with affine and curved parts); a separate instantiation so
// that the common single-kind launches carry no list look-up in their prefetch address chains (measured: 0.182 -> 0.211 ms with it)
// SPLIT: split-vector input (multi-rank applies without L-vector copies); its own instantiation, because a per-lane choice of the
// base pointer costs the affine kernels the two registers they have left at three waves per SIMD
template <int PT, int MODE, bool AFFINE, bool CPLX = false, bool LIST = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * ((AFFINE && !CPLX) ? kAffWaves : kResWaves), 1) void dense_apply_resident_kernel(const DenseArgs a, const int rows) {
  static_assert(!SPLIT || (!CPLX && !LIST), "split vectors: whole real operators");
  static_assert(!CPLX || (MODE == MODE_CURLMASS && PT <= 3), "complex form: curl-curl + mass blocks");
  constexpr int NW = (AFFINE && !CPLX) ? kAffWaves : kResWaves;
  using M = ModeTraits<MODE>;
  using F0 = FieldTraits<MODE, 0>;
  using F1 = FieldTraits<MODE, 1>;
  constexpr int NCT = M::NCT, KPMAX = 4 * PT, NF = F0::NF, S = ResidentStride<PT>::S;
  [[maybe_unused]] constexpr int NQ0 = F0::NC == 3 ? 6 : (F0::NC == 2 ? 3 : 1), NQ1 = F1::NC == 3 ? 6 : (F1::NC == 2 ? 3 : 1);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // (the wave number as a scalar: block numbers and with them the bases of the index / orientation / q-data / E-vector rows
  // live in SGPRs, the lanes add a 32-bit offset -- as 64-bit per-lane pointers they cost the registers whose spill put a
  // scratch reload + vmcnt(0) between the index prefetch and the matrix products)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, kq = lane >> 4;
  constexpr int KP = 4 * PT;  // the host pads every element block to 4 PT dof slots
  const int woff = (a.Q4 + 31) / 32 * 32;  // relative quadrature weights first (a multiple of 32 doubles: L keeps its banks)
  const double *wl = smem;
  double *L = smem + woff;
  double *sm = L + (size_t)rows * S + (size_t)wave * KPMAX * 64;
  {
    for (int i = tid; i < woff; i += 64 * NW) smem[i] = (a.wrel && i < a.Q4) ? a.wrel[i] : 0.0;
    const int n = rows * S;
    for (int i = tid; i < n; i += 64 * NW) L[i] = a.L[i];
  }
  __syncthreads();
  const size_t cs = (size_t)a.Q4 * kEB;
  const int ngroups = a.Q4 / 4;
  // work units: element blocks, or (CPLX) half blocks -- unit w is columns 8 (w & 1) .. + 7 of block w >> 1, and lane (kq, j)
  // works on element column 8 (w & 1) + (j & 7), part j >> 3
  const int nunits = (CPLX ? 2 : 1) * (LIST ? a.nblist : a.nb);  // (LIST: the blocks of one kind of a mesh with affine and curved ones)
  const int j8 = j & 7;
  auto ublock = [&](const int w) {
    const int bw = CPLX ? w >> 1 : w;
    return (size_t)(LIST ? __builtin_amdgcn_readfirstlane(a.blist[bw]) : bw);
  };
  auto ucol = [&](const int w) { return CPLX ? 8 * (w & 1) + j8 : j; };          // element column in the block's arrays
  auto ulane = [&](const int w) { return CPLX ? kq * 16 + 8 * (w & 1) + j8 : lane; };  // position in a [.][64] row
  const double *xsel = (CPLX && (j >> 3)) ? a.x1 : a.x;
  [[maybe_unused]] const double *xgh = SPLIT ? dense_ghosts(a) : xsel;
  const double part_sign = (j >> 3) ? 1.0 : -1.0;  // real part: - A_i x_i, imaginary part: + A_i x_r

  // Software pipeline over the wave's blocks (up to 16 dof slots per lane: beyond that the extra registers spill): the index
  // words of the next block are requested while the current one is in the matrix cores, its x values once the products
  // are done (the q-data registers are free then) so that they fly during the E^T stores; a block starts with its
  // operands in registers instead of two dependent memory round trips.
  constexpr bool PREFETCH_IDX = KPMAX <= 16, PREFETCH_X = KPMAX <= 12;
  // XEARLY (experiment builds, -DPA_DENSE_XEARLY=1): x of the next block requested BEFORE the matrix products of the current one,
  // the index words two blocks ahead.  Measured in round 5 on 280k tets (profiles/r05_dense_ab.log): at twelve waves per
  // workgroup the 24 extra live registers spill (0.190 / 0.286 ms against 0.183 / 0.265), at eight waves it equals the
  // one-block distance at twelve (0.183 / 0.268): the kernel is not waiting for x.  Default off.
  constexpr bool XEARLY = PREFETCH_X && AFFINE && !CPLX && (PA_DENSE_XEARLY != 0);
  int sgn[PREFETCH_IDX ? KPMAX : 1];
  double un[PREFETCH_X ? KPMAX : 1];
  // likewise the first chunk of q-data and the curl-orientation words of the next block (values the compiler cannot hold in
  // vector registers go to the accumulation registers, which two waves per SIMD leave free)
  double qdn[4][6];
  double qdi[CPLX ? 2 : 1][6];  // complex form: point-0 D of the imaginary operator {mass, curl-curl}
  unsigned con[PREFETCH_X ? KPMAX / 2 : 1];  // (two 16-bit words per register: a.co2)
  // first chunk of field 0 of unit w, or for an affine block the point-0 values of both fields (raw, scaled at use)
  auto request_qd = [&](const int w, double (&out)[4][6]) {
    const double *qb = a.qdata + ublock(w) * a.ncq * cs;
    if (AFFINE) {
      const int col = ucol(w);
#pragma unroll
      for (int k = 0; k < NQ0; k++) out[0][k] = qb[k * cs + col];
      if (NF == 2) {
#pragma unroll
        for (int k = 0; k < NQ1; k++) out[1][k] = qb[(NQ0 + k) * cs + col];
      }
      if (CPLX) {
        const double *qi = a.qdata_i + ublock(w) * a.ncq_i * cs + col;
#pragma unroll
        for (int k = 0; k < 6; k++) {
          qdi[0][k] = a.qi_mass >= 0 ? qi[(a.qi_mass + k) * cs] : 0.0;
          qdi[CPLX ? 1 : 0][k] = a.qi_curl >= 0 ? qi[(a.qi_curl + k) * cs] : 0.0;
        }
      }
    } else {
      load_qd<MODE, 0>(qb + ulane(w), cs, ngroups, out);
    }
  };
  auto gather_x = [&](const int (&sg_)[PREFETCH_IDX ? KPMAX : 1], double (&out)[PREFETCH_X ? KPMAX : 1]) {
#pragma unroll
    for (int s = 0; s < (PREFETCH_X ? KPMAX : 1); s++) {
      out[s] = 0.0;
      if (s < KP) {
        const int sg = sg_[s];
        const int d = sg >= 0 ? sg : -1 - sg;
        const double xv = (d & kEssBit) ? 0.0 : (SPLIT ? dense_xbase(a, xsel, xgh, d & ~kEssBit) : xsel)[d & ~kEssBit];
        out[s] = sg >= 0 ? xv : -xv;
      }
    }
  };
  if (PREFETCH_IDX) {
    const int b0 = blockIdx.x * NW + wave, w0 = b0 < nunits ? b0 : 0;
    // (uniform row base + 32-bit lane offset: the address stays "SGPR pair + VGPR", nothing per lane to keep across the loop)
    const int32_t *idx0 = a.idx + ublock(w0) * KP * 64;
    const unsigned l0 = (unsigned)ulane(w0);
#pragma unroll
    for (int s = 0; s < KPMAX; s++) sgn[s] = (s < KP) ? idx0[s * 64u + l0] : 0;
    if (PREFETCH_X) {
      gather_x(sgn, un);
      request_qd(w0, qdn);
      if (a.co2) {
#pragma unroll
        for (int s = 0; s < KPMAX / 2; s++) con[s] = (a.co2 + ublock(w0) * (KP / 2) * 64)[s * 64u + (unsigned)ulane(w0)];
      }
      if (XEARLY) {  // index words of the second block of this wave (clamped): its x is requested inside the first iteration
        const int b1 = b0 + (int)gridDim.x * NW, w1 = b1 < nunits ? b1 : w0;
        const int32_t *idx1 = a.idx + ublock(w1) * KP * 64;
        const unsigned l1 = (unsigned)ulane(w1);
#pragma unroll
        for (int s = 0; s < KPMAX; s++) sgn[s] = (s < KP) ? idx1[s * 64u + l1] : 0;
      }
    }
  }
  for (int b = blockIdx.x * NW + wave; b < nunits; b += gridDim.x * NW) {
    const size_t bb = ublock(b);  // the element block of this unit
    const int gln = ulane(b);     // this lane's position in its [.][64] rows
    // ---- E
    double u[KPMAX];
#pragma unroll
    for (int s = 0; s < KPMAX; s++) {
      u[s] = 0.0;
      if (s < KP) {
        if (PREFETCH_X) {
          u[s] = un[s];
          continue;
        }
        const int sg = PREFETCH_IDX ? sgn[s] : a.idx[bb * KP * 64 + s * 64 + gln];
        const int d = sg >= 0 ? sg : -1 - sg;
#ifdef PA_ABLATION
        const double xv = (a.dbg & 1) ? (double)d : ((d & kEssBit) ? 0.0 : (SPLIT ? dense_xbase(a, xsel, xgh, d & ~kEssBit) : xsel)[d & ~kEssBit]);
#else
        const double xv = (d & kEssBit) ? 0.0 : (SPLIT ? dense_xbase(a, xsel, xgh, d & ~kEssBit) : xsel)[d & ~kEssBit];
#endif
        u[s] = sg >= 0 ? xv : -xv;
      }
    }
#ifdef PA_ABLATION
    const double *q = a.qdata + ((a.dbg & 2) ? (size_t)0 : bb * a.ncq * cs) + lane;
#else
    const double *q = a.qdata + bb * a.ncq * cs + gln;
#endif
    [[maybe_unused]] const double *qim = (CPLX && !AFFINE) ? a.qdata_i + bb * a.ncq_i * cs + gln : nullptr;
    double qd[4][6];
    if (AFFINE) {  // qdn[0], qdn[1]: point-0 values of the two fields, kept until the next request
      if (!PREFETCH_X) request_qd(b, qdn);
    } else if (!PREFETCH_X) {
      load_qd<MODE, 0>(q, cs, ngroups, qd);
    } else {
#pragma unroll
      for (int gl = 0; gl < 4; gl++)
#pragma unroll
        for (int k = 0; k < NQ0; k++) qd[gl][k] = qdn[gl][k];
    }
#ifdef PA_ABLATION
    const bool co = a.co2 && !(a.dbg & 16);
#else
    const bool co = a.co2 != nullptr;
#endif
    // this block's curl-orientation words, two per register, kept for E^T: re-reading them from memory there put twelve
    // load -> wait -> store round trips behind the x gather of the NEXT block (one in-order counter for loads and stores),
    // i.e. the whole latency the software pipeline exists to hide was paid once per block (round 5, found in the ISA;
    // stage ablation: 39 of 139 us)
    unsigned cpk[KPMAX / 2];
    if (co) {
#pragma unroll
      for (int s = 0; s < KPMAX / 2; s++) cpk[s] = PREFETCH_X ? con[s] : (a.co2 + bb * (KP / 2) * 64)[s * 64u + (unsigned)gln];
#pragma unroll
      for (int s = 0; s < KPMAX; s++)
        if (s < KP) sm[s * 64 + lane] = u[s];
      wave_sync();
#pragma unroll
      for (int s = 0; s < KPMAX; s++) {
        if (s < KP) {
          const unsigned c = cpk[s >> 1];
          const int dof = 4 * s + kq;
          // out-of-range neighbours have a zero coefficient: clamp the address instead of branching
          const double lo = sm[max(dof - 1, 0) * 16 + j];
          const double hi = sm[min(dof + 1, 4 * KP - 1) * 16 + j];
          u[s] = co_field_pk(c, s & 1, 0) * lo + co_field_pk(c, s & 1, 1) * u[s] + co_field_pk(c, s & 1, 2) * hi;
        }
      }
      wave_sync();
    }
    double4_t yacc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; pt++) yacc[pt] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int bn = b + (int)gridDim.x * NW;
    const int bnc = bn < nunits ? bn : b;  // the next unit (clamped on the last one)
    // complex form: the D of the imaginary operator is needed until the products below are done; its next request overwrites it
    double qdic[CPLX ? 2 : 1][6];
    if (CPLX) {
#pragma unroll
      for (int f = 0; f < (CPLX ? 2 : 1); f++)
#pragma unroll
        for (int k = 0; k < 6; k++) qdic[f][k] = qdi[f][k];
    }
    // XEARLY: the x values of the NEXT block are requested here, before the matrix products (they fly during the products instead of
    // during the short E^T stage; its index words were requested one block earlier still, so the index prefetch below runs two
    // blocks ahead)
    if (XEARLY) gather_x(sgn, un);
    if (PREFETCH_IDX) {  // its index words
      // (the lane offset through an opaque copy: hoisted out of the loop as 64-bit per-lane pointers these two addresses were
      // spilled, and their scratch reload put a vmcnt(0) between the index prefetch and the matrix products)
      int ln = ulane(bnc);
      asm volatile("" : "+v"(ln));
      const int bn2 = bn + (int)gridDim.x * NW;
      const int bic = XEARLY ? (bn2 < nunits ? bn2 : bnc) : bnc;  // whose index words: two blocks ahead with XEARLY
      const int32_t *idxn = a.idx + ublock(bic) * KP * 64 + (XEARLY ? ulane(bic) : ln);
#pragma unroll
      for (int s = 0; s < KPMAX; s++)
        if (s < KP) sgn[s] = idxn[s * 64];
      if (PREFETCH_X && co) {
        const uint32_t *con_n = a.co2 + ublock(bnc) * (KP / 2) * 64 + ln;
#pragma unroll
        for (int s = 0; s < KPMAX / 2; s++) con[s] = con_n[s * 64];
      }
    }

#ifdef PA_ABLATION
    const int nch_eff = (a.dbg & 4) ? 0 : a.nch;
    if (a.dbg & 4) {
#pragma unroll
      for (int pt = 0; pt < PT; pt++)
        yacc[pt] = double4_t{u[(4 * pt) % KPMAX] + qd[0][0], u[(4 * pt + 1) % KPMAX], u[(4 * pt + 2) % KPMAX], u[(4 * pt + 3) % KPMAX]};
    }
#else
    const int nch_eff = a.nch;
#endif
    // (the table addresses of this lane re-derived from an opaque copy of the lane id: kept across the E / E^T stages they are
    // what the register allocator spills, and a scratch reload here waits -- one in-order counter -- for the index prefetch)
    int lo_ = lane;
    asm volatile("" : "+v"(lo_));
    const double *Lf = L + (lo_ & 15) * S + (lo_ >> 4);  // forward operand of this lane:   row 16 t + i, column 4 s + kq
    const double *Lb = L + (lo_ >> 4) * S + (lo_ & 15);  // transposed operand of this lane: row 4 pi + kq, column 16 pt + i
    for (int c = 0; c < nch_eff; c++) {
      const int r0 = c * NCT * 16;
      if (AFFINE) {
        const double *wc = wl + 16 * c + kq;
        resident_field<PT, MODE, 0>(Lf + r0 * S, Lb + r0 * S, KP, u, qdn, yacc, wc, CPLX ? qdic[0] : nullptr, part_sign);
        if (NF == 2)
          resident_field<PT, MODE, 1>(Lf + (r0 + 16 * F0::NC) * S, Lb + (r0 + 16 * F0::NC) * S, KP, u, qdn + 1, yacc, wc,
                                      CPLX ? qdic[CPLX ? 1 : 0] : nullptr, part_sign);
      } else if (NF == 1) {
        double qn[4][6];
        if (c + 1 < a.nch) load_qd<MODE, 0>(q + 16 * (c + 1) * kEB, cs, ngroups - 4 * (c + 1), qn);
        resident_field<PT, MODE, 0>(Lf + r0 * S, Lb + r0 * S, KP, u, qd, yacc);
        if (c + 1 < a.nch) {
#pragma unroll
          for (int gl = 0; gl < 4; gl++)
#pragma unroll
            for (int k = 0; k < NQ0; k++) qd[gl][k] = qn[gl][k];
        }
      } else if (CPLX) {  // curved blocks, complex form
        double qn[4][6], qi[4][6];
        load_qd6(qim + 16 * c * kEB, a.qi_mass, cs, ngroups - 4 * c, qi);
        load_qd<MODE, 1>(q + NQ0 * cs + 16 * c * kEB, cs, ngroups - 4 * c, qn);
        resident_field<PT, MODE, 0>(Lf + r0 * S, Lb + r0 * S, KP, u, qd, yacc, nullptr, nullptr, part_sign, qi);
        load_qd6(qim + 16 * c * kEB, a.qi_curl, cs, ngroups - 4 * c, qi);
        if (c + 1 < a.nch) load_qd<MODE, 0>(q + 16 * (c + 1) * kEB, cs, ngroups - 4 * (c + 1), qd);
        resident_field<PT, MODE, 1>(Lf + (r0 + 16 * F0::NC) * S, Lb + (r0 + 16 * F0::NC) * S, KP, u, qn, yacc, nullptr, nullptr,
                                    part_sign, qi);
      } else {
        double qn[4][6];
        load_qd<MODE, 1>(q + NQ0 * cs + 16 * c * kEB, cs, ngroups - 4 * c, qn);
        resident_field<PT, MODE, 0>(Lf + r0 * S, Lb + r0 * S, KP, u, qd, yacc);
        if (c + 1 < a.nch) load_qd<MODE, 0>(q + 16 * (c + 1) * kEB, cs, ngroups - 4 * (c + 1), qd);
        resident_field<PT, MODE, 1>(Lf + (r0 + 16 * F0::NC) * S, Lb + (r0 + 16 * F0::NC) * S, KP, u, qn, yacc);
      }
    }

    if (PREFETCH_X) {  // the first q-data chunk of the next block (and its x values, unless they were requested before the products)
      if (!XEARLY) gather_x(sgn, un);
      request_qd(bnc, qdn);
    }

    // ---- E^T, first half
#ifdef PA_ABLATION
    double *ye = a.ye + ((a.dbg & 8) ? (size_t)(blockIdx.x * NW + wave) : bb) * KP * 64;
#else
    double *ye = ((CPLX && (j >> 3)) ? a.ye1 : a.ye) + bb * KP * 64;
#endif
    // position of dof slot s of this lane: base + s * stride (ye_pos; two integers instead of the expression per slot)
    const int ye_stride = a.ye_rows ? 4 : 64;
    double *yel = ye + (a.ye_rows ? (gln & 15) * (4 * KP) + (gln >> 4) : gln);
    if (co) {
#pragma unroll
      for (int s = 0; s < KPMAX; s++)
        if (s < KP) sm[s * 64 + lane] = yacc[s >> 2][s & 3];
      wave_sync();
#pragma unroll
      for (int s = 0; s < KPMAX; s++) {
        if (s < KP) {
          const int dof = 4 * s + kq;
          const unsigned cm = cpk[s >> 1];
          yel[s * ye_stride] = co_field_pk(cm, s & 1, 1) * yacc[s >> 2][s & 3] + co_field_pk(cm, s & 1, 3) * sm[max(dof - 1, 0) * 16 + j] +
                             co_field_pk(cm, s & 1, 4) * sm[min(dof + 1, 4 * KP - 1) * 16 + j];
        }
      }
      wave_sync();
    } else {
#pragma unroll
      for (int s = 0; s < KPMAX; s++)
        if (s < KP) yel[s * ye_stride] = yacc[s >> 2][s & 3];
    }
  }
}

// packed D (set-up): one thread per point; columns of D through the reference QFunction arithmetic
template <int MODE>
__global__ void dense_qdata_kernel(const DenseArgs a, double *__restrict__ qd) {
  using F0 = FieldTraits<MODE, 0>;
  using F1 [[maybe_unused]] = FieldTraits<MODE, 1>;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / a.Q);
  if (e >= a.ne) return;
  const int q = (int)(gid - (long long)e * a.Q);
  const size_t cs = (size_t)a.Qpad * kEB;
  const double *g = a.geom + ((size_t)(e / kEB) * 11 * a.Qpad + q) * kEB + (e % kEB);
  double *out = qd + ((size_t)(e / kEB) * a.ncq * a.Q4 + q) * kEB + (e % kEB);
  const size_t os = (size_t)a.Q4 * kEB;
  double adj[9];
  for (int k = 0; k < 9; k++) adj[k] = g[(2 + k) * cs];
  const double wdetJ = g[cs];
  const int attr = (int)g[0];
  int o = 0;
  if (MODE == MODE_DIFFMASS && a.contra) {
    // f_apply_l2mass_33 (l2mass_33_qf.h:10-42) on the tables of an H(div) element, divergence in the place of the values and values
    // in the place of the gradient: c qw^2 / (w detJ) (second context), then the H(div) mass w detJ Jl^T C Jl (first context)
    out[0] = a.c0.mat[coeff_index(a.c0, attr)] * a.qw[q] * a.qw[q] / wdetJ;
    double Jl[9], Cm[9], Mx[9];
    coeff_unpack3(a.c1, attr, Cm);
    adjJt33(adj, Jl);
    for (int col = 0; col < 3; col++)
      mult_AtBCx33(Jl, Cm, Jl, col == 0 ? 1.0 : 0.0, col == 1 ? 1.0 : 0.0, col == 2 ? 1.0 : 0.0, wdetJ, Mx[0 + 3 * col],
                   Mx[1 + 3 * col], Mx[2 + 3 * col]);
    out[1 * os] = Mx[0], out[2 * os] = 0.5 * (Mx[3] + Mx[1]), out[3 * os] = 0.5 * (Mx[6] + Mx[2]);
    out[4 * os] = Mx[4], out[5 * os] = 0.5 * (Mx[7] + Mx[5]), out[6 * os] = Mx[8];
    return;
  }
  auto field = [&](auto tag) {
    constexpr int F = decltype(tag)::value;
    constexpr int NC = FieldTraits<MODE, F>::NC;
    if (NC == 3) {
      double Mx[9];
      for (int col = 0; col < 3; col++) {
        double v[3] = {col == 0 ? 1.0 : 0.0, col == 1 ? 1.0 : 0.0, col == 2 ? 1.0 : 0.0};
        dense_D_field<MODE, F>(a, wdetJ, adj, attr, v);
        Mx[0 + 3 * col] = v[0], Mx[1 + 3 * col] = v[1], Mx[2 + 3 * col] = v[2];
      }
      out[(o + 0) * os] = Mx[0];
      out[(o + 1) * os] = 0.5 * (Mx[3] + Mx[1]);
      out[(o + 2) * os] = 0.5 * (Mx[6] + Mx[2]);
      out[(o + 3) * os] = Mx[4];
      out[(o + 4) * os] = 0.5 * (Mx[7] + Mx[5]);
      out[(o + 5) * os] = Mx[8];
      o += 6;
    } else {
      double v[1] = {1.0};
      dense_D_field<MODE, F>(a, wdetJ, adj, attr, v);
      out[o * os] = v[0];
      o += 1;
    }
  };
  field(std::integral_constant<int, 0>{});
  if (F0::NF == 2) field(std::integral_constant<int, 1>{});
}

// Line elements: geometry data {attr, w |J|, J / |J|^2 (SDIM rows)} (geom_21_qf.h:9-30, geom_31_qf.h:9-31) and the scalar D of
// every form on them:  w detJ a^T C a with a = adj(J)^T / detJ (hcurl_21_qf.h:10-29, hcurl_31_qf.h; C is SDIM x SDIM), c w detJ
// (h1_1_qf.h, first half of hcurlmass_21 / _31)
template <int SDIM>
__global__ void geom_dense1_kernel(const int ne, const int Q, const int Qpad, const int npe, const int32_t *__restrict__ off,
                                   const double *__restrict__ nodes, const int32_t *__restrict__ attr,
                                   const double *__restrict__ grad, const double *__restrict__ w, double *__restrict__ geom) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  double J[SDIM] = {};
  for (int n = 0; n < npe; n++) {
    const double gq = grad[(size_t)q * npe + n];
    const double *x = nodes + (size_t)off[(size_t)e * npe + n] * SDIM;
    for (int i = 0; i < SDIM; i++) J[i] += gq * x[i];
  }
  double d2 = 0.0;
  for (int i = 0; i < SDIM; i++) d2 += J[i] * J[i];
  const double d = sqrt(d2);
  const size_t cs = (size_t)Qpad * kEB;
  double *g = geom + ((size_t)(e / kEB) * (2 + SDIM) * Qpad + q) * kEB + (e % kEB);
  g[0] = (double)attr[e];
  g[cs] = w[q] * d;
  for (int i = 0; i < SDIM; i++) g[(2 + i) * cs] = (J[i] / d) / d;
}

template <int MODE, int SDIM>
__global__ void dense_qdata1_kernel(const DenseArgs a, double *__restrict__ qd) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / a.Q);
  if (e >= a.ne) return;
  const int q = (int)(gid - (long long)e * a.Q);
  const size_t cs = (size_t)a.Qpad * kEB, os = (size_t)a.Q4 * kEB;
  const double *g = a.geom + ((size_t)(e / kEB) * (2 + SDIM) * a.Qpad + q) * kEB + (e % kEB);
  double *out = qd + ((size_t)(e / kEB) * a.ncq * a.Q4 + q) * kEB + (e % kEB);
  const int attr = (int)g[0];
  const double wdetJ = g[cs];
  int o = 0;
  if (MODE == MODE_DIFFMASS1 && a.contra) {
    // f_apply_l2mass_21 | _31 (l2mass_21_qf.h:10-40): H(div) mass on the values (below, first context), c qw^2 / (w detJ) on the
    // divergence (second context)
    out[os] = a.c1.mat[coeff_index(a.c1, attr)] * a.qw[q] * a.qw[q] / wdetJ;
  } else if (MODE == MODE_MASS || MODE == MODE_DIFFMASS1) {
    out[(o++) * os] = a.c0.mat[coeff_index(a.c0, attr)] * wdetJ;
  }
  if (MODE != MODE_MASS) {  // MultAtBCx21 / MultAtBCx31 with x = 1 (utils_21_qf.h, utils_31_qf.h:41-59)
    const CoeffDev &cc = (MODE == MODE_DIFFMASS1 && !a.contra) ? a.c1 : a.c0;
    const double *C = cc.mat + SDIM * SDIM * coeff_index(cc, attr);
    double av[SDIM], s = 0.0;
    for (int i = 0; i < SDIM; i++) av[i] = g[(2 + i) * cs];
    if (a.contra) {  // AdjJt21 / AdjJt31 of the stored vector (utils_21_qf.h:19-28, utils_31_qf.h:19-31): a / |a|
      double n2 = 0.0;
      for (int i = 0; i < SDIM; i++) n2 += av[i] * av[i];
      const double d = sqrt(n2);
      for (int i = 0; i < SDIM; i++) av[i] /= d;
    }
    for (int i = 0; i < SDIM; i++) {
      double z = 0.0;
      for (int j = 0; j < SDIM; j++) z += C[i + SDIM * j] * av[j];
      s += av[i] * z;
    }
    out[o * os] = wdetJ * s;
  }
}

// 2-D elements, in the plane (6-row geometry data {attr, w detJ, adj(J)^T/detJ 2x2}, 2x2 materials) or on the boundary of a 3-D
// mesh (BDR: 8 rows with the 3x2 adj(J)^T/detJ, 3x3 materials): packed D per field in the order {values, derivatives} --
//   symmetric 2x2 {00, 01, 11} = w detJ A^T C A    hcurl_22_qf.h:10-30, hcurl_32_qf.h:10-30 (ND mass, H1 diffusion)
//   c qw^2 / (w detJ)                              l2_1_qf.h:10-24, second half of hdivmass_22 / _32 (scalar curl)
//   c w detJ                                       h1_1_qf.h, first half of hcurlmass_22 / _32 (H1 mass)
template <int MODE, int NROWS>
__global__ void dense_qdata2_kernel(const DenseArgs a, double *__restrict__ qd) {
  constexpr bool BDR = NROWS == 8;  // (NROWS = 11: 3-D geometry data, scalar forms only -- div-div on Raviart-Thomas elements)
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / a.Q);
  if (e >= a.ne) return;
  const int q = (int)(gid - (long long)e * a.Q);
  const size_t cs = (size_t)a.Qpad * kEB, os = (size_t)a.Q4 * kEB;
  const double *g = a.geom + ((size_t)(e / kEB) * NROWS * a.Qpad + q) * kEB + (e % kEB);
  double *out = qd + ((size_t)(e / kEB) * a.ncq * a.Q4 + q) * kEB + (e % kEB);
  const int attr = (int)g[0];
  const double wdetJ = g[cs];
  int o = 0;
  auto block22 = [&](const CoeffDev &cc) {
    double Mx[4];
    if (BDR) {  // MultAtBCx32(adjJt, coeff, adjJt, e_col) * wdetJ (utils_32_qf.h:53-72)
      double A[6], C[9];
      for (int k = 0; k < 6; k++) A[k] = g[(2 + k) * cs];
      if (a.contra) {  // f_apply_hdiv_32 (hdiv_32_qf.h:10-31) first takes AdjJt32 of the stored matrix (utils_32_qf.h:23-40)
        const double E = A[0] * A[0] + A[1] * A[1] + A[2] * A[2], G = A[3] * A[3] + A[4] * A[4] + A[5] * A[5];
        const double F = A[0] * A[3] + A[1] * A[4] + A[2] * A[5], d = sqrt(E * G - F * F);
        const double B6[6] = {(G * A[0] - F * A[3]) / d, (G * A[1] - F * A[4]) / d, (G * A[2] - F * A[5]) / d,
                              (E * A[3] - F * A[0]) / d, (E * A[4] - F * A[1]) / d, (E * A[5] - F * A[2]) / d};
        for (int k = 0; k < 6; k++) A[k] = B6[k];
      }
      coeff_unpack3(cc, attr, C);
      for (int col = 0; col < 2; col++) {
        const double x0 = col == 0 ? 1.0 : 0.0, x1 = col == 1 ? 1.0 : 0.0;
        const double y0 = A[0] * x0 + A[3] * x1, y1 = A[1] * x0 + A[4] * x1, t = A[2] * x0 + A[5] * x1;
        const double z0 = C[0] * y0 + C[3] * y1 + C[6] * t, z1 = C[1] * y0 + C[4] * y1 + C[7] * t,
                     z2 = C[2] * y0 + C[5] * y1 + C[8] * t;
        Mx[0 + 2 * col] = wdetJ * (A[0] * z0 + A[1] * z1 + A[2] * z2);
        Mx[1 + 2 * col] = wdetJ * (A[3] * z0 + A[4] * z1 + A[5] * z2);
      }
    } else {  // MultAtBCx22 (utils_22_qf.h); f_apply_hdiv_22 (hdiv_22_qf.h:10-30) first takes AdjJt22 of the stored matrix
      const double G4[4] = {g[2 * cs], g[3 * cs], g[4 * cs], g[5 * cs]};
      const double A[4] = {a.contra ? G4[3] : G4[0], a.contra ? -G4[2] : G4[1], a.contra ? -G4[1] : G4[2],
                           a.contra ? G4[0] : G4[3]};
      const double *C = cc.mat + 4 * coeff_index(cc, attr);  // CoeffUnpack2, column-major
      for (int col = 0; col < 2; col++) {
        const double x0 = col == 0 ? 1.0 : 0.0, x1 = col == 1 ? 1.0 : 0.0;
        const double y0 = A[0] * x0 + A[2] * x1, y1 = A[1] * x0 + A[3] * x1;
        const double z0 = C[0] * y0 + C[2] * y1, z1 = C[1] * y0 + C[3] * y1;
        Mx[0 + 2 * col] = wdetJ * (A[0] * z0 + A[1] * z1);
        Mx[1 + 2 * col] = wdetJ * (A[2] * z0 + A[3] * z1);
      }
    }
    out[o * os] = Mx[0], out[(o + 1) * os] = 0.5 * (Mx[1] + Mx[2]), out[(o + 2) * os] = Mx[3];
    o += 3;
  };
  if (MODE == MODE_MASS || MODE == MODE_DIFFMASS2) {  // values of a scalar field first
    out[o * os] = a.c0.mat[coeff_index(a.c0, attr)] * wdetJ;
    o += 1;
  }
  if (MODE == MODE_VMASS2 || MODE == MODE_CURLMASS2 || MODE == MODE_DIFF2) block22(a.c0);
  if (MODE == MODE_DIFFMASS2) block22(a.c1);
  if (MODE == MODE_CURL2 || MODE == MODE_CURLMASS2) {
    const CoeffDev &cc = (MODE == MODE_CURL2) ? a.c0 : a.c1;
    const double w = a.qw[q];
    out[o * os] = cc.mat[coeff_index(cc, attr)] * w * w / wdetJ;
  }
}

// Diagonal from the packed D (any mode that has q-data): d_e[j] = sum_q b_j^T D b_j, pushed through the
// transpose of the unsigned restriction (see dense_diag_kernel)
template <int MODE>
__global__ void dense_diag_qd_kernel(const DenseArgs a, const int32_t *__restrict__ off, const int8_t *__restrict__ cor,
                                     const double *__restrict__ interp, const double *__restrict__ deriv,
                                     double *__restrict__ diag) {
  using M = ModeTraits<MODE>;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / a.P);
  if (e >= a.ne) return;
  const int jd = (int)(gid - (long long)e * a.P);
  const size_t os = (size_t)a.Q4 * kEB;
  const double *qd = a.qdata + ((size_t)(e / kEB) * a.ncq * a.Q4) * kEB + (e % kEB);
  double d = 0.0;
  for (int q = 0; q < a.Q; q++) {
    int o = 0;
    auto part = [&](const double *tab, int nc) {
      double v[3], w[3];
      for (int k = 0; k < nc; k++) v[k] = w[k] = tab[((size_t)k * a.Q + q) * a.P + jd];
      double m[6];
      const int nq = nc == 3 ? 6 : (nc == 2 ? 3 : 1);
      for (int k = 0; k < nq; k++) m[k] = qd[(size_t)(o + k) * os + (size_t)q * kEB];
      if (nc == 3) sym_mv(m, v[0], v[1], v[2], w[0], w[1], w[2]);
      else if (nc == 2) w[0] = m[0] * v[0] + m[1] * v[1], w[1] = m[1] * v[0] + m[2] * v[1];
      else w[0] = m[0] * v[0];
      for (int k = 0; k < nc; k++) d += v[k] * w[k];
      o += nq;
    };
    if (M::NCI > 0) part(interp, M::NCI);
    if (M::NCD > 0) part(deriv, M::NCD);
  }
  diag[gid] = d;  // element diagonal; dense_diag_slot_kernel pushes it through the transposed unsigned restriction
}

// split-vector launches: the 3-D forms a multi-rank solve applies (curl-curl, vector mass, both, diffusion), PT <= 3, one kind of block
static bool resident_split_mode(int mode) { return mode == MODE_CURL || mode == MODE_VMASS || mode == MODE_CURLMASS || mode == MODE_DIFF; }
template <int PT>
void launch_resident_split(const DenseSub &ds, const DenseArgs &a, hipStream_t s) {
  const int rows = ds.L_rows;
  const bool affine = a.affine != nullptr;
  const int nw = affine ? kAffWaves : kResWaves;
  const size_t shm = sizeof(double) * ((size_t)(a.Q4 + 31) / 32 * 32 + (size_t)rows * ResidentStride<PT>::S +
                                      (ds.d_co ? (size_t)nw * 4 * PT * 64 : 0));
  if (ds.nb == 0) return;
  const int grid = std::min((ds.nb + nw - 1) / nw, ds.num_cu);
  switch (ds.mode) {
#define PA_RES_SPLIT_CASE(MODE)                                                                                          \
  case MODE: {                                                                                                           \
    static std::atomic<bool> attr_set{false};                                                                            \
    if (!attr_set) {                                                                                                     \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, false, false, false, true>,         \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                               \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, true, false, false, true>,          \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                               \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    if (affine)                                                                                                          \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, true, false, false, true>), dim3(grid), dim3(64 * nw), shm, s, a, rows); \
    else                                                                                                                 \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, false, false, false, true>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows); \
  } break;
    PA_RES_SPLIT_CASE(MODE_CURL)
    PA_RES_SPLIT_CASE(MODE_VMASS)
    PA_RES_SPLIT_CASE(MODE_CURLMASS)
    PA_RES_SPLIT_CASE(MODE_DIFF)
#undef PA_RES_SPLIT_CASE
    default: throw Error("no split-vector form of this dense block");
  }
  PA_HIP(hipGetLastError());
}

template <int PT>
void launch_resident_pt(const DenseSub &ds, const DenseArgs &a, hipStream_t s) {
  const int rows = ds.L_rows;
  if (ds.d_blist[0] && !a.blist) {  // affine and curved blocks: one launch each on its list (they write disjoint E-vector blocks)
    DenseArgs aa = a, ag = a;
    aa.blist = ds.d_blist[0], aa.nblist = ds.n_blist[0];
    ag.blist = ds.d_blist[1], ag.nblist = ds.n_blist[1], ag.affine = nullptr;
    launch_resident_pt<PT>(ds, aa, s);
    launch_resident_pt<PT>(ds, ag, s);
    return;
  }
  const bool affine = a.affine && PT <= 3;  // (make_dense_sub: larger blocks have no registers for the values kept across the block)
  const int nw = affine ? kAffWaves : kResWaves;
  const size_t shm = sizeof(double) * ((size_t)(a.Q4 + 31) / 32 * 32 + (size_t)rows * ResidentStride<PT>::S +
                                      (ds.d_co ? (size_t)nw * 4 * PT * 64 : 0));
  const int nblocks = a.blist ? a.nblist : ds.nb;
  if (nblocks == 0) return;
  int grid = (nblocks + nw - 1) / nw;
  if (grid > ds.num_cu) grid = ds.num_cu;
  switch (ds.mode) {
#define PA_RES_CASE(MODE)                                                                                \
  case MODE: {                                                                                           \
    static std::atomic<bool> attr_set{false}; /* idempotent set-up; rank threads may race here */ \
    if (!attr_set) {                                                                                     \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, false>,             \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, (PT <= 3)>,         \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, false, false, (PT <= 3)>, \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE, (PT <= 3), false, (PT <= 3)>, \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    if (a.blist && affine)                                                                               \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, (PT <= 3), false, (PT <= 3)>), dim3(grid), dim3(64 * nw), shm, s, a, rows); \
    else if (a.blist)                                                                                    \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, false, false, (PT <= 3)>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows); \
    else if (affine)                                                                                     \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, (PT <= 3)>), dim3(grid), dim3(64 * nw), shm, s, a, rows); \
    else                                                                                                 \
      hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE, false>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows); \
  } break;
    PA_RES_CASE(MODE_CURL)
    PA_RES_CASE(MODE_VMASS)
    PA_RES_CASE(MODE_CURLMASS)
    PA_RES_CASE(MODE_DIFF)
    PA_RES_CASE(MODE_DIFFMASS)
    PA_RES_CASE(MODE_MASS)
    PA_RES_CASE(MODE_CURL2)
    PA_RES_CASE(MODE_VMASS2)
    PA_RES_CASE(MODE_CURLMASS2)
    PA_RES_CASE(MODE_DIFF2)
    PA_RES_CASE(MODE_DIFFMASS2)
    PA_RES_CASE(MODE_VMASS1)
    PA_RES_CASE(MODE_DIFF1)
    PA_RES_CASE(MODE_DIFFMASS1)
#undef PA_RES_CASE
  }
}

template <int PT>
void launch_pt(const DenseSub &ds, const DenseArgs &a, hipStream_t s) {
  const dim3 grid((ds.nb + kDenseWaves - 1) / kDenseWaves), block(kDenseThreads);
  // LDS: the fragments of one field of one chunk (forward and transposed share the buffer) and, for
  // the curl-oriented restriction, the per-wave neighbour exchange
  const size_t shm = sizeof(double) * ((size_t)3 * PT * kDenseThreads + (ds.d_co ? (size_t)kDenseWaves * 4 * PT * 64 : 0));
  switch (ds.mode) {
#define PA_DENSE_CASE(MODE)                                                                              \
  case MODE: {                                                                                           \
    static std::atomic<bool> attr_set{false}; /* idempotent set-up; rank threads may race here */ \
    if (!attr_set) {                                                                                     \
      PA_HIP(hipFuncSetAttribute((const void *)dense_apply_kernel<PT, MODE>,                             \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));               \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL((dense_apply_kernel<PT, MODE>), grid, block, shm, s, a);                          \
  } break;
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

// fem/qfunctions/22/geom_22_qf.h:9-30: {attr, w detJ, adj(J)^T / detJ} with adj(J)^T = {J3, -J2, -J1, J0}
__global__ void geom_dense2_kernel(const int ne, const int Q, const int Qpad, const int npe,
                                   const int32_t *__restrict__ node_off, const double *__restrict__ nodes,
                                   const int32_t *__restrict__ attr, const double *__restrict__ grad,
                                   const double *__restrict__ w, double *__restrict__ geom) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  double J[4] = {0, 0, 0, 0};
  for (int n = 0; n < npe; n++) {
    const int id = node_off[(size_t)e * npe + n];
    const double d0 = grad[((size_t)0 * Q + q) * npe + n], d1 = grad[((size_t)1 * Q + q) * npe + n];
    for (int c = 0; c < 2; c++) {
      const double X = nodes[2 * (size_t)id + c];
      J[c + 0] += X * d0;
      J[c + 2] += X * d1;
    }
  }
  const double det = J[0] * J[3] - J[1] * J[2];
  double *g = geom + ((size_t)(e / kEB) * 6 * Qpad + q) * kEB + (e % kEB);
  const size_t cs = (size_t)Qpad * kEB;
  g[0] = (double)attr[e];
  g[cs] = w[q] * det;
  g[2 * cs] = J[3] / det, g[3 * cs] = -J[2] / det, g[4 * cs] = -J[1] / det, g[5 * cs] = J[0] / det;
}

// fem/qfunctions/32/geom_32_qf.h:9-33 with utils_32_qf.h:23-40: J is 3x2, detJ = sqrt(E G - F^2)
__global__ void geom_dense32_kernel(const int ne, const int Q, const int Qpad, const int npe,
                                    const int32_t *__restrict__ node_off, const double *__restrict__ nodes,
                                    const int32_t *__restrict__ attr, const double *__restrict__ grad,
                                    const double *__restrict__ w, double *__restrict__ geom) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  double J[6] = {0, 0, 0, 0, 0, 0};
  for (int n = 0; n < npe; n++) {
    const int id = node_off[(size_t)e * npe + n];
    const double d0 = grad[((size_t)0 * Q + q) * npe + n], d1 = grad[((size_t)1 * Q + q) * npe + n];
    for (int c = 0; c < 3; c++) {
      const double X = nodes[3 * (size_t)id + c];
      J[c + 0] += X * d0;
      J[c + 3] += X * d1;
    }
  }
  const double E = J[0] * J[0] + J[1] * J[1] + J[2] * J[2], G = J[3] * J[3] + J[4] * J[4] + J[5] * J[5];
  const double F = J[0] * J[3] + J[1] * J[4] + J[2] * J[5];
  const double d = sqrt(E * G - F * F);
  double *g = geom + ((size_t)(e / kEB) * 8 * Qpad + q) * kEB + (e % kEB);
  const size_t cs = (size_t)Qpad * kEB;
  g[0] = (double)attr[e];
  g[cs] = w[q] * d;
  for (int k = 0; k < 3; k++) {
    g[(2 + k) * cs] = (G * J[k] - F * J[3 + k]) / d / d;
    g[(5 + k) * cs] = (E * J[3 + k] - F * J[k]) / d / d;
  }
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
  diag[gid] = d;  // element diagonal; dense_diag_slot_kernel pushes it through the transposed unsigned restriction
}

// Second half of the diagonal: the E-vector entry of (element e, local dof j) in the block layout of the apply kernels.
// Plain / oriented restriction: d_e[j]; curl-oriented: (|T_e|^T d_e)[j] = |T[j][j]| d[j] + |T[j+1][j]| d[j+1] + |T[j-1][j]| d[j-1].
// The gather that follows applies the orientation sign of the entry, which a diagonal does not have: it is cancelled here.
// (Fixed summation order: the diagonal -- and with it every smoother built on it -- is identical from run to run.)
__global__ void dense_diag_slot_kernel(const int ne, const int P, const int KP, const int8_t *__restrict__ cor,
                                       const int32_t *__restrict__ idx, const double *__restrict__ de,
                                       double *__restrict__ ye, const int ye_rows) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / P);
  if (e >= ne) return;
  const int j = (int)(gid - (long long)e * P);
  const double *d = de + (size_t)e * P;
  double v = d[j];
  if (cor) {
    const int8_t *t = cor + 3 * ((size_t)e * P + j);
    v = fabs((double)t[1]) * d[j];
    if (j + 1 < P) v += fabs((double)t[3 + 0]) * d[j + 1];
    if (j > 0) v += fabs((double)t[-3 + 2]) * d[j - 1];
  }
  const size_t pos = ((size_t)(e / kEB) * 4 * KP + j) * kEB + (e % kEB);  // (the index table keeps the [dof][element] rows)
  const size_t blk = (size_t)(e / kEB) * 4 * KP * kEB;
  ye[ye_rows ? blk + (size_t)(e % kEB) * 4 * KP + j : pos] = idx[pos] < 0 ? -v : v;
}

// Blocks whose packed D depends on the point through the quadrature weight only (constant Jacobian and attribute: straight-sided
// simplices): w_0 D_q = w_q D_0 to 1e-13 -- the Jacobian of a straight-sided higher-order element is a sum over its nodes with
// cancellation of order (domain size / element size) roundings; linear elements are constant to the bit.  One thread per
// (block, element slot).
__global__ void dense_affine_kernel(const int nb, const int ncq, const int Q, const int Q4, const double *__restrict__ qd,
                                    const double *__restrict__ wq, unsigned int *__restrict__ not_affine) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nb * kEB) return;
  const int b = gid / kEB, j = gid % kEB;
  const size_t cs = (size_t)Q4 * kEB;
  const double *e = qd + (size_t)b * ncq * cs + j;
  double scale = 0.0;
  for (int k = 0; k < ncq; k++) scale = fmax(scale, fabs(e[k * cs]));
  bool ok = true;
  for (int k = 0; k < ncq && ok; k++) {
    const double d0 = e[k * cs];
    for (int q = 1; q < Q; q++)
      if (fabs(e[k * cs + (size_t)q * kEB] * wq[0] - d0 * wq[q]) > 1e-13 * scale * fabs(wq[q])) {
        ok = false;
        break;
      }
  }
  if (!ok) atomicOr(&not_affine[b], 1u);
}

DenseArgs make_args(const DenseSub &ds) {
  DenseArgs a;
  a.ne = ds.ne, a.nb = ds.nb, a.P = ds.P, a.Q = ds.Q, a.Qpad = ds.Qpad, a.nch = ds.nch, a.KP = ds.KP;
  a.idx = ds.d_idx, a.co = ds.d_co, a.co2 = ds.d_co2, a.geom = ds.geom->d_geom, a.qw = ds.geom->d_qw, a.Tf = ds.d_Tf, a.Tt = ds.d_Tt;
  a.contra = ds.contra ? 1 : 0;
  a.L = ds.d_L, a.qdata = ds.d_qdata, a.ncq = ds.ncq, a.Q4 = (ds.Q + 3) / 4 * 4;
  a.affine = ds.d_affine, a.wrel = ds.d_wrel;
  a.x1 = nullptr, a.ye1 = nullptr, a.qdata_i = nullptr, a.ncq_i = 0, a.qi_mass = a.qi_curl = -1;
  a.blist = nullptr, a.nblist = 0;
  a.dbg = 0;
#ifdef PA_ABLATION
  a.dbg = getenv("PA_DBG") ? atoi(getenv("PA_DBG")) : 0;
#endif
  a.x = nullptr, a.ye = ds.d_ye, a.ye_rows = ds.ye_rows ? 1 : 0;
  a.nsplit = 0x7fffffff, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr;
  a.c0 = ds.c0.dev(), a.c1 = ds.c1.dev();
  return a;
}

}  // namespace

void launch_geom_dense(const pa_mesh_dense_desc &mesh, Geom &g, hipStream_t s) {
  const int ne = mesh.num_elem, npe = mesh.nodes_per_elem, Q = mesh.num_qpts;
  const int dim = mesh.dim == 0 ? 3 : mesh.dim;
  const int sdim = mesh.space_dim == 0 ? dim : mesh.space_dim;
  PA_REQUIRE(dim >= 1 && dim <= 3, "element dimension must be 1, 2 or 3");
  PA_REQUIRE(dim == 1 ? (sdim == 2 || sdim == 3) : (sdim == dim || (dim == 2 && sdim == 3)),
             "space dimension: that of the element, 3 for boundary elements, 2 or 3 for line elements");
  PA_REQUIRE(ne > 0 && npe > 0 && Q > 0 && mesh.num_nodes > 0, "empty mesh description");
  PA_REQUIRE(mesh.node_offsets && mesh.nodes && mesh.attr && mesh.mesh_grad && mesh.qweight, "null mesh array");
  for (size_t i = 0; i < (size_t)ne * npe; i++)
    PA_REQUIRE(mesh.node_offsets[i] >= 0 && mesh.node_offsets[i] < mesh.num_nodes, "mesh node id out of range");
  for (int e = 0; e < ne; e++) PA_REQUIRE(mesh.attr[e] >= 1, "element attributes are 1-based");
  int32_t *d_off = dev_upload(mesh.node_offsets, (size_t)ne * npe, s);
  double *d_nodes = dev_upload(mesh.nodes, (size_t)mesh.num_nodes * sdim, s);
  int32_t *d_attr = dev_upload(mesh.attr, (size_t)ne, s);
  double *d_grad = dev_upload(mesh.mesh_grad, (size_t)dim * Q * npe, s);
  double *d_w = dev_upload(mesh.qweight, (size_t)Q, s);
  g.ne = ne, g.q1d = 0, g.Q = Q, g.eb = kEB, g.Qpad = (Q + 15) / 16 * 16;
  g.dim = dim, g.sdim = sdim, g.nrows = dim == 1 ? 2 + sdim : (dim == 3 ? 11 : (sdim == 3 ? 8 : 6));
  g.d_qw = dev_upload(mesh.qweight, (size_t)Q, s);
  g.wq.assign(mesh.qweight, mesh.qweight + Q);
  const size_t nb = (size_t)(ne + kEB - 1) / kEB, count = nb * g.nrows * g.Qpad * kEB;
  g.d_geom = dev_alloc<double>(count);
  PA_HIP(hipMemsetAsync(g.d_geom, 0, sizeof(double) * count, s));
  const long long n = (long long)ne * Q;
  const int bs = 256;
  if (dim == 1 && sdim == 3)
    hipLaunchKernelGGL(geom_dense1_kernel<3>, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                       d_nodes, d_attr, d_grad, d_w, g.d_geom);
  else if (dim == 1)
    hipLaunchKernelGGL(geom_dense1_kernel<2>, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                       d_nodes, d_attr, d_grad, d_w, g.d_geom);
  else if (dim == 2 && sdim == 3)
    hipLaunchKernelGGL(geom_dense32_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                       d_nodes, d_attr, d_grad, d_w, g.d_geom);
  else if (dim == 2)
    hipLaunchKernelGGL(geom_dense2_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                       d_nodes, d_attr, d_grad, d_w, g.d_geom);
  else
    hipLaunchKernelGGL(geom_dense_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne, Q, g.Qpad, npe, d_off,
                       d_nodes, d_attr, d_grad, d_w, g.d_geom);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  hipFree(d_off), hipFree(d_nodes), hipFree(d_attr), hipFree(d_grad), hipFree(d_w);
}

DenseSub *make_dense_sub(pa_geom *geom, const pa_restriction_desc &r, const pa_dense_basis_desc &b, int qf,
                         const void *ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops, int height, bool contra) {
  PA_REQUIRE(geom && geom->eb == kEB, "geometry data must come from pa_geom_create_dense");
  if (b.fe_type == PA_FE_HDIV) {
    // H(div) elements run on the arithmetic of an existing mode: the tables change places, the D of the values takes the
    // contravariant map (`contra`: AdjJt of the stored adj(J)^T / detJ, what f_apply_hdiv_* do first).
    const int d = 10 * geom->sdim + geom->dim;
    const uint32_t t_ops = trial_ops & ~(uint32_t)PA_EVAL_WEIGHT, s_ops = test_ops & ~(uint32_t)PA_EVAL_WEIGHT;
    pa_dense_basis_desc alias = b;
    if (qf == PA_QF_L2_1) {
      // div-div (fem/integ/divdiv.cpp: Div | Weight, f_apply_l2_1 for single-component elements): the scalar-derivative
      // arithmetic of the 2-D curl-curl, c qw^2 / (w detJ), with the divergence table [Q][P] in the place of the curl table
      PA_REQUIRE((d == 33 || d == 22 || d == 32) && t_ops == PA_EVAL_DIV && s_ops == PA_EVAL_DIV && b.deriv,
                 "H(div) elements: the div-div operator takes the divergence table with Div (| Weight)");
      alias.fe_type = PA_FE_HCURL, alias.interp = nullptr;
      return make_dense_sub(geom, r, alias, qf, ctx, ctx_size, PA_EVAL_CURL, PA_EVAL_CURL, height);
    }
    const int q_mass = d == 33 ? PA_QF_HDIV_33 : d == 22 ? PA_QF_HDIV_22 : d == 32 ? PA_QF_HDIV_32 : d == 21 ? PA_QF_HDIV_21 : PA_QF_HDIV_31;
    const int q_pair = d == 33   ? PA_QF_L2MASS_33
                       : d == 22 ? PA_QF_L2MASS_22
                       : d == 32 ? PA_QF_L2MASS_32
                       : d == 21 ? PA_QF_L2MASS_21
                                 : PA_QF_L2MASS_31;
    if (qf == q_mass) {
      PA_REQUIRE(t_ops == PA_EVAL_INTERP && s_ops == PA_EVAL_INTERP && b.interp, "H(div) mass: Interp on the value table");
      if (d == 33) {
        // (fem/integ/vecfemass.cpp with an RT space: Interp + f_apply_hdiv_33) the arithmetic of the curl-curl operator with the
        // value table in the place of the curl table
        alias.fe_type = PA_FE_HCURL, alias.deriv = b.interp, alias.interp = nullptr;
        return make_dense_sub(geom, r, alias, qf, ctx, ctx_size, PA_EVAL_CURL, PA_EVAL_CURL, height);
      }
      // plane, boundary and line elements (f_apply_hdiv_22 | _32 | _21 | _31): the vector mass of the geometry with the
      // contravariant map in the place of adjJt
      alias.fe_type = PA_FE_HCURL;
      const int q_cov = d == 22 ? PA_QF_HCURL_22 : d == 32 ? PA_QF_HCURL_32 : d == 21 ? PA_QF_HCURL_21 : PA_QF_HCURL_31;
      return make_dense_sub(geom, r, alias, q_cov, ctx, ctx_size, PA_EVAL_INTERP, PA_EVAL_INTERP, height, true);
    }
    PA_REQUIRE(qf == q_pair, "H(div) elements: the mass operator (Interp, hdiv_*), div-div (Div, l2_1) and div-div + mass "
                             "(Interp | Div, l2mass_*) of the geometry's dimensions are supported");
    // DivDivMassIntegrator (fem/integ/divdivmass.cpp: Interp | Div | Weight, f_apply_l2mass_*; pair context: mass first)
    PA_REQUIRE(t_ops == (PA_EVAL_INTERP | PA_EVAL_DIV) && s_ops == t_ops && b.interp && b.deriv,
               "div-div + mass: value and divergence tables with Interp | Div (| Weight)");
    if (d == 22 || d == 32) {  // the plane / boundary curl-curl + mass: two values, one scalar derivative
      alias.fe_type = PA_FE_HCURL;
      return make_dense_sub(geom, r, alias, d == 22 ? PA_QF_HDIVMASS_22 : PA_QF_HDIVMASS_32, ctx, ctx_size,
                            PA_EVAL_INTERP | PA_EVAL_CURL, PA_EVAL_INTERP | PA_EVAL_CURL, height, true);
    }
    alias.fe_type = PA_FE_H1;
    if (d == 33) {  // diffusion + mass with the roles exchanged: one "value" (the divergence), three "derivatives" (the values)
      alias.interp = b.deriv, alias.deriv = b.interp;
      return make_dense_sub(geom, r, alias, PA_QF_HCURLMASS_33, ctx, ctx_size, PA_EVAL_INTERP | PA_EVAL_GRAD,
                            PA_EVAL_INTERP | PA_EVAL_GRAD, height, true);
    }
    // line elements: one value, one derivative -- the tables stay where they are
    return make_dense_sub(geom, r, alias, d == 21 ? PA_QF_HCURLMASS_21 : PA_QF_HCURLMASS_31, ctx, ctx_size,
                          PA_EVAL_INTERP | PA_EVAL_GRAD, PA_EVAL_INTERP | PA_EVAL_GRAD, height, true);
  }
  PA_REQUIRE(b.fe_type == PA_FE_H1 || b.fe_type == PA_FE_HCURL, "unknown element type");
  PA_REQUIRE(b.num_dofs > 0 && b.num_qpts == geom->Q, "basis and geometry data disagree on the quadrature rule");
  PA_REQUIRE(r.num_elem == geom->ne && r.elem_size == b.num_dofs, "restriction does not match mesh / basis");
  PA_REQUIRE(r.lsize == height && r.offsets, "restriction L-vector size does not match the operator");
  PA_REQUIRE(!(r.orients && r.curl_orients), "restriction is either oriented or curl-oriented");
  PA_REQUIRE(r.lsize < (1 << 29), "too many local dofs for the index encoding");
  const int P = b.num_dofs, Q = b.num_qpts, ne = r.num_elem;
  const int dim = geom->dim;
  const int sdim = geom->sdim;
  const int mode = mode_of(b.fe_type, qf, dim, sdim);
  int nci, ncd;
  mode_comps(mode, nci, ncd);
  const int nct = nci + ncd;
  PA_REQUIRE(nci == 0 || b.interp, "interp table missing");
  PA_REQUIRE(ncd == 0 || b.deriv, "curl / gradient table missing");
  {
    const uint32_t want_i = nci ? PA_EVAL_INTERP : 0u;
    const uint32_t want_d = ncd ? (b.fe_type == PA_FE_HCURL ? PA_EVAL_CURL : PA_EVAL_GRAD) : 0u;
    // 2-D curl-curl adds the Weight input (integ/curlcurl.cpp:65-68): accepted, the weights live in the geometry data
    const uint32_t got = trial_ops & ~(uint32_t)PA_EVAL_WEIGHT;
    // (the reference adds Weight to the trial side only, integ/curlcurl.cpp:62-68)
    PA_REQUIRE(got == (want_i | want_d) && (test_ops & ~(uint32_t)PA_EVAL_WEIGHT) == got, "eval modes do not match the QFunction");
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
  ds->ne = ne, ds->nb = (ne + kEB - 1) / kEB, ds->lsize = r.lsize, ds->KP = 4 * PT, ds->PT = PT;
  ds->qf = qf, ds->mode = mode, ds->trial_ops = trial_ops, ds->test_ops = test_ops, ds->contra = contra;
  const int KP = ds->KP, nb = ds->nb, nch = ds->nch;

  // ---- E: block-transposed index (+ packed tridiagonal rows)
  const size_t nslot = (size_t)nb * KP * 64;
  std::vector<int32_t> idx(nslot, kEssBit);
  std::vector<uint16_t> co(r.curl_orients ? nslot : 0, 0u);
  for (int e = 0; e < ne; e++) {
    for (int d = 0; d < P; d++) {
      const int32_t off = r.offsets[(size_t)e * P + d];
      PA_REQUIRE(off >= 0 && off < r.lsize, "restriction offset out of range");
      const size_t pos = ((size_t)(e / kEB) * 4 * KP + d) * kEB + (e % kEB);
      idx[pos] = (r.orients && r.orients[(size_t)e * P + d]) ? -1 - off : off;
      if (r.curl_orients) {
        const int8_t *t = r.curl_orients + 3 * ((size_t)e * P + d);
        const int8_t up = d > 0 ? t[-3 + 2] : 0, dn = d + 1 < P ? t[3 + 0] : 0;  // T[d-1][d], T[d+1][d]
        // MFEM's ND face transformations only contain -1, 0, 1 (the reference stores them as int8,
        // restriction.cpp:318-336); two bits per entry
        for (int8_t v : {t[0], t[1], t[2]}) PA_REQUIRE(v >= -1 && v <= 1, "curl-orientation entries must be -1, 0 or 1");
        co[pos] = (uint16_t)((t[0] & 3) | ((t[1] & 3) << 2) | ((t[2] & 3) << 4) | ((up & 3) << 6) | ((dn & 3) << 8));
      }
    }
  }
  ds->d_idx = dev_upload(idx.data(), nslot);
  if (r.curl_orients) {
    ds->d_co = dev_upload(co.data(), nslot);
    // the resident kernel takes the words of slots 2 s, 2 s + 1 of a lane together (half the loads and registers)
    std::vector<uint32_t> co2(nslot / 2);
    for (size_t b = 0; b < (size_t)nb; b++)
      for (int s2 = 0; s2 < KP / 2; s2++)
        for (int l = 0; l < 64; l++)
          co2[(b * (KP / 2) + s2) * 64 + l] = (uint32_t)co[(b * KP + 2 * s2) * 64 + l] | ((uint32_t)co[(b * KP + 2 * s2 + 1) * 64 + l] << 16);
    ds->d_co2 = dev_upload(co2.data(), co2.size());
  }
  ds->h_co = co;
  // transpose map for the gather form of E^T (counting sort by dof, element order preserved)
  {
    // Which rows the E-vector gets (DenseArgs::ye_rows) is decided by what the GATHER does with them, MEASURED here on blocks large
    // enough for it to matter: a mesh numbered with locality in the [dof] rows (a Kuhn-split cube: consecutive elements share
    // entities at the same local index, and neighbouring waves of the gather re-read each other's rows from L2) keeps them -- there
    // the rows by element only cost the apply kernel its coalesced stores (+3 us of 186) -- an unstructured mesh (config 3's) reads
    // one 64-byte sector per 8-byte entry from the [dof] rows and a third of that from the rows by element (gather of 2.18M order-3
    // dofs: 64 -> 37 us).  The sums do not depend on the choice (same copies, same order): timing noise cannot change a result.
    // PALACE_AMD_DENSE_ELAYOUT=rows | block overrides (read at every creation: A / B runs in one process); the run form of the
    // gather is written for the [dof] rows.
    std::vector<int32_t> tptr((size_t)r.lsize + 1, 0), tent((size_t)ne * P), ted((size_t)ne * P);
    for (size_t k = 0; k < (size_t)ne * P; k++) tptr[(size_t)r.offsets[k] + 1]++;
    for (int d = 0; d < r.lsize; d++) tptr[d + 1] += tptr[d];
    {
      std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
      for (int e = 0; e < ne; e++)
        for (int d = 0; d < P; d++) ted[fill[r.offsets[(size_t)e * P + d]]++] = e * P + d;
    }
    auto fill_tent = [&](bool rows) {
      for (size_t k = 0; k < (size_t)ne * P; k++) {
        const int e = ted[k] / P, d = ted[k] % P;
        const int32_t pos = (int32_t)ye_pos_host(rows, KP, e, d);
        const bool flip = r.orients && r.orients[(size_t)e * P + d];
        tent[k] = flip ? -1 - pos : pos;
      }
    };
    ds->d_tptr = dev_upload(tptr.data(), tptr.size());
    ds->d_ye = dev_alloc<double>(nslot);
    const char *gl = getenv("PALACE_AMD_DENSE_ELAYOUT"), *gg = getenv("PALACE_AMD_DENSE_GATHER");
    const bool runs_form = gg && std::string(gg) == "runs";
    if (gl && std::string(gl) == "rows") ds->ye_rows = true;
    else if ((gl && std::string(gl) == "block") || runs_form || r.lsize < (1 << 17)) ds->ye_rows = false;
    else {
      PA_HIP(hipMemset(ds->d_ye, 0, nslot * sizeof(double)));
      double t[2];
      for (int rows = 0; rows < 2; rows++) {
        fill_tent(rows != 0);
        int32_t *d_t = dev_upload(tent.data(), tent.size());
        t[rows] = time_dense_gather(*ds, d_t);
        (void)hipFree(d_t);
      }
      ds->ye_rows = t[1] < 0.85 * t[0];
      if (getenv("PALACE_AMD_DENSE_VERBOSE"))
        std::fprintf(stderr, "[palace_amd] dense block %d elements x %d dofs: gather %.1f us by dof rows, %.1f us by element rows -> %s\n", ne,
                     P, 1e3 * t[0], 1e3 * t[1], ds->ye_rows ? "element rows" : "dof rows");
    }
    fill_tent(ds->ye_rows);
    ds->d_tent = dev_upload(tent.data(), tent.size());
    // a block that touches few of the dofs (the surface terms of a driven problem: a few thousand boundary faces in a space of
    // millions) adds its result through the list of the rows it has: its gather was a pass over the whole vector, 58 times per
    // FGMRES iteration of config 3 (round 5: 3.7 % of that solve's device time)
    if ((size_t)ne * P * 4 < (size_t)r.lsize) {
      std::vector<int32_t> rows;
      for (int d = 0; d < r.lsize; d++)
        if (tptr[d + 1] > tptr[d]) rows.push_back(d);
      ds->n_rows = (int)rows.size();
      ds->d_rows = dev_upload(rows.data(), std::max<size_t>(rows.size(), 1));
    }
    // ... and, on request (PALACE_AMD_DENSE_GATHER=runs), its run form.  Measured on 279 936 order-3 tetrahedra, alternating on one box:
    // curl-curl 0.1857 / 0.1866 ms with the CSR form against 0.1839 / 0.1846 ms by runs, K + M 0.2631 / 0.2633 against 0.2613 / 0.2618
    // (-1 %: the gather is bound by its scattered 8-byte E-vector reads, not by the position words), and SLOWER on small blocks
    // (14 362 cubic H1 tetrahedra: 0.076 against 0.031 ms, four dofs per thread leave 65 workgroups) -- the CSR form stays the default.
    static const bool runs = getenv("PALACE_AMD_DENSE_GATHER") && std::string(getenv("PALACE_AMD_DENSE_GATHER")) == "runs";
    if (runs) {
      std::vector<uint32_t> code, rpos;
      std::vector<streamhost::RunHdr> hdr;
      streamhost::build_runs_dense(ne, P, KP, r.lsize, r.offsets, r.orients, code, hdr, rpos);
      const std::vector<streamhost::RunChunk> ch = streamhost::run_chunks(code);
      ds->d_rchunk = dev_upload(reinterpret_cast<const uint32_t *>(ch.data()), 4 * ch.size());
      ds->d_rhdr = dev_upload(reinterpret_cast<const int32_t *>(hdr.data()), 2 * hdr.size());
      ds->d_rpos_run = dev_upload(rpos.data(), std::max<size_t>(rpos.size(), 1));
    }
  }
  ds->h_idx = std::move(idx);

  // ---- B: MFMA A-operand fragments (see the header comment for the row order)
  auto tab = [&](int cidx, int q, int p) -> double {
    if (q >= Q || p >= P) return 0.0;
    return cidx < nci ? b.interp[((size_t)cidx * Q + q) * P + p] : b.deriv[((size_t)(cidx - nci) * Q + q) * P + p];
  };
  std::vector<double> Tf((size_t)nch * KP * nct * 64), Tt((size_t)nch * 4 * nct * PT * 64);
  {
    const int nf = (nci > 0 ? 1 : 0) + (ncd > 0 ? 1 : 0);
    size_t of = 0, ot = 0;
    for (int c = 0; c < nch; c++)
      for (int f = 0; f < nf; f++) {
        const int nc = (nf == 2) ? (f == 0 ? nci : ncd) : nct, cb = (nf == 2 && f == 1) ? nci : 0;
        for (int s = 0; s < KP; s++)
          for (int t = 0; t < nc; t++)
            for (int lane = 0; lane < 64; lane++) {
              const int i = lane & 15, kq = lane >> 4;
              const int pi = 4 * t + (i >> 2), gl = pi / nc, k = pi % nc;
              Tf[of++] = tab(cb + k, 16 * c + 4 * gl + (i & 3), 4 * s + kq);
            }
        for (int pi = 0; pi < 4 * nc; pi++)
          for (int pt = 0; pt < PT; pt++)
            for (int lane = 0; lane < 64; lane++) {
              const int i = lane & 15, kq = lane >> 4;
              const int gl = pi / nc, k = pi % nc;
              Tt[ot++] = tab(cb + k, 16 * c + 4 * gl + kq, 16 * pt + i);
            }
      }
  }
  ds->d_Tf = dev_upload(Tf.data(), Tf.size());
  ds->d_Tt = dev_upload(Tt.data(), Tt.size());
  {  // position-weighted checksums of the two tables (two sub-operators on "the same basis" are compared through them)
    auto chk = [&](const double *t, int nc) {
      double c = 0.0;
      if (t)
        for (size_t i = 0; i < (size_t)nc * Q * P; i++) c += t[i] * (double)(1 + i % 1021);
      return c;
    };
    ds->chk_interp = chk(b.interp, nci), ds->chk_deriv = chk(b.deriv, ncd);
  }
  // plain copies for the diagonal kernel
  if (nci) ds->d_interp = dev_upload(b.interp, (size_t)nci * Q * P);
  if (ncd) ds->d_deriv = dev_upload(b.deriv, (size_t)ncd * Q * P);
  ds->d_off = dev_upload(r.offsets, (size_t)ne * P);
  ds->h_off.assign(r.offsets, r.offsets + (size_t)ne * P);
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
      if (contra) {  // l2mass_33: the 3 x 3 mass coefficient comes first, then the scalar one of the divergence
        parse_coeff(ctx, ctx_size, 3, ds->c1, 0);
        parse_coeff(ctx, ctx_size, 1, ds->c0, ds->c1.slots);
        break;
      }
      parse_coeff(ctx, ctx_size, 1, ds->c0, 0);
      parse_coeff(ctx, ctx_size, 3, ds->c1, ds->c0.slots);
      break;
    case MODE_MASS:
    case MODE_CURL2:
      parse_coeff(ctx, ctx_size, 1, ds->c0, 0);
      break;
    case MODE_VMASS2:
    case MODE_DIFF2:
      parse_coeff(ctx, ctx_size, sdim == 3 ? 3 : 2, ds->c0, 0);  // boundary elements take the 3x3 material
      break;
    case MODE_CURLMASS2:
      parse_coeff(ctx, ctx_size, sdim == 3 ? 3 : 2, ds->c0, 0);
      parse_coeff(ctx, ctx_size, 1, ds->c1, ds->c0.slots);
      break;
    case MODE_DIFFMASS2:
    case MODE_DIFFMASS1:
      if (mode == MODE_DIFFMASS1 && contra) {  // l2mass_21 | _31: the mass coefficient matrix first, then the scalar one
        parse_coeff(ctx, ctx_size, sdim == 3 ? 3 : 2, ds->c0, 0);
        parse_coeff(ctx, ctx_size, 1, ds->c1, ds->c0.slots);
        break;
      }
      parse_coeff(ctx, ctx_size, 1, ds->c0, 0);
      parse_coeff(ctx, ctx_size, sdim == 3 ? 3 : 2, ds->c1, ds->c0.slots);
      break;
    case MODE_VMASS1:
    case MODE_DIFF1:
      parse_coeff(ctx, ctx_size, sdim == 3 ? 3 : 2, ds->c0, 0);
      break;
  }

  // ---- fast path: tables resident in LDS (one copy, rows in tile order, stride S) + packed D.
  // Needs symmetric coefficients and tables that fit; PALACE_AMD_DENSE=staged keeps the general kernel.
  {
    auto is_sym = [](const CoeffHost &c) {
      if (c.dim == 2) {
        for (size_t k = 0; k + 4 <= c.mat.size(); k += 4)
          if (c.mat[k + 1] != c.mat[k + 2]) return false;
        return true;
      }
      if (c.dim != 3) return true;
      for (size_t k = 0; k + 9 <= c.mat.size(); k += 9)
        if (c.mat[k + 1] != c.mat[k + 3] || c.mat[k + 2] != c.mat[k + 6] || c.mat[k + 5] != c.mat[k + 7]) return false;
      return true;
    };
    const int S = 16 * PT + ((PT & 1) ? 2 : 18);
    const int rows = nch * nct * 16;
    const size_t lds = sizeof(double) * ((size_t)rows * S + (r.curl_orients ? (size_t)8 * 4 * PT * 64 : 0));
    const char *force = getenv("PALACE_AMD_DENSE");
    const bool staged = force && std::string(force) == "staged";
    if (!staged && lds <= 150 * 1024 && is_sym(ds->c0) && (ds->c1.mat.empty() || is_sym(ds->c1))) {
      const int nf = (nci > 0 ? 1 : 0) + (ncd > 0 ? 1 : 0);
      std::vector<double> L((size_t)rows * S, 0.0);
      for (int c = 0; c < nch; c++)
        for (int f = 0; f < nf; f++) {
          const int nc = (nf == 2) ? (f == 0 ? nci : ncd) : nct, cb = (nf == 2 && f == 1) ? nci : 0;
          for (int row = 0; row < 16 * nc; row++) {
            const int pi = row >> 2, kqp = row & 3, gl = pi / nc, k = pi % nc;
            for (int d = 0; d < P; d++)
              L[((size_t)(c * nct + cb) * 16 + row) * S + d] = tab(cb + k, 16 * c + 4 * gl + kqp, d);
          }
        }
      ds->d_L = dev_upload(L.data(), L.size());
      ds->L_rows = rows;
      ds->ncq = (nci == 3 ? 6 : (nci == 2 ? 3 : nci)) + (ncd == 3 ? 6 : (ncd == 2 ? 3 : ncd));
      const size_t nq = (size_t)nb * ds->ncq * ((Q + 3) / 4 * 4) * kEB;
      ds->d_qdata = dev_alloc<double>(nq);
      PA_HIP(hipMemset(ds->d_qdata, 0, sizeof(double) * nq));
      hipDeviceProp_t prop;
      int dev = 0;
      PA_HIP(hipGetDevice(&dev));
      PA_HIP(hipGetDeviceProperties(&prop, dev));
      ds->num_cu = prop.multiProcessorCount;
      DenseArgs a = make_args(*ds);
      const long long n = (long long)ne * Q;
      const dim3 grid((unsigned)((n + 255) / 256)), block(256);
      if (dim == 1) {
        switch (mode) {
#define PA_QD1_CASE(MODE)                                                                          \
  case MODE:                                                                                       \
    if (sdim == 3)                                                                                 \
      hipLaunchKernelGGL((dense_qdata1_kernel<MODE, 3>), grid, block, 0, nullptr, a, ds->d_qdata); \
    else                                                                                           \
      hipLaunchKernelGGL((dense_qdata1_kernel<MODE, 2>), grid, block, 0, nullptr, a, ds->d_qdata); \
    break;
          PA_QD1_CASE(MODE_MASS)
          PA_QD1_CASE(MODE_VMASS1)
          PA_QD1_CASE(MODE_DIFF1)
          PA_QD1_CASE(MODE_DIFFMASS1)
#undef PA_QD1_CASE
        }
      } else if (dim == 3 && mode == MODE_CURL2) {
        hipLaunchKernelGGL((dense_qdata2_kernel<MODE_CURL2, 11>), grid, block, 0, nullptr, a, ds->d_qdata);
      } else if (dim == 2) {
        switch (mode) {
#define PA_QD2_CASE(MODE)                                                                          \
  case MODE:                                                                                       \
    if (sdim == 3)                                                                                 \
      hipLaunchKernelGGL((dense_qdata2_kernel<MODE, 8>), grid, block, 0, nullptr, a, ds->d_qdata); \
    else                                                                                           \
      hipLaunchKernelGGL((dense_qdata2_kernel<MODE, 6>), grid, block, 0, nullptr, a, ds->d_qdata); \
    break;
          PA_QD2_CASE(MODE_CURL2)
          PA_QD2_CASE(MODE_VMASS2)
          PA_QD2_CASE(MODE_CURLMASS2)
          PA_QD2_CASE(MODE_MASS)
          PA_QD2_CASE(MODE_DIFF2)
          PA_QD2_CASE(MODE_DIFFMASS2)
#undef PA_QD2_CASE
        }
      } else
      switch (mode) {
#define PA_QD_CASE(MODE) \
  case MODE: hipLaunchKernelGGL((dense_qdata_kernel<MODE>), grid, block, 0, nullptr, a, ds->d_qdata); break;
        PA_QD_CASE(MODE_CURL)
        PA_QD_CASE(MODE_VMASS)
        PA_QD_CASE(MODE_CURLMASS)
        PA_QD_CASE(MODE_DIFF)
        PA_QD_CASE(MODE_DIFFMASS)
        PA_QD_CASE(MODE_MASS)
#undef PA_QD_CASE
      }
      PA_HIP(hipGetLastError());
      PA_HIP(hipStreamSynchronize(nullptr));
      // affine blocks read one point of q-data (PALACE_AMD_DENSE_AFFINE=0: every block reads all of it)
      const char *aff_env = getenv("PALACE_AMD_DENSE_AFFINE");
      const size_t lds_aff = sizeof(double) * ((size_t)((Q + 3) / 4 * 4 + 31) / 32 * 32 + (size_t)rows * S +
                                               (r.curl_orients ? (size_t)kAffWaves * 4 * PT * 64 : 0));
      if (!(aff_env && aff_env[0] == '0') && PT <= 3 && lds_aff <= 160 * 1024 && (int)geom->wq.size() == Q && Q > 1 &&
          geom->wq[0] != 0.0) {
        const int Q4 = (Q + 3) / 4 * 4;
        unsigned int *d_na = dev_alloc<unsigned int>((size_t)nb);
        PA_HIP(hipMemset(d_na, 0, sizeof(unsigned int) * nb));
        hipLaunchKernelGGL(dense_affine_kernel, dim3((unsigned)((nb * kEB + 255) / 256)), dim3(256), 0, nullptr, nb, ds->ncq, Q,
                           Q4, ds->d_qdata, geom->d_qw, d_na);
        PA_HIP(hipGetLastError());
        std::vector<unsigned int> na((size_t)nb);
        PA_HIP(hipMemcpy(na.data(), d_na, sizeof(unsigned int) * nb, hipMemcpyDeviceToHost));
        hipFree(d_na);
        std::vector<uint8_t> flag((size_t)nb);
        for (int i = 0; i < nb; i++) flag[i] = na[i] ? 0 : 1, ds->n_affine += flag[i];
        if (ds->n_affine < nb && ds->n_affine * 4 >= nb) {  // a mesh with curved parts: the affine blocks get the affine kernel
          std::vector<int32_t> lists[2];
          for (int i = 0; i < nb; i++) lists[flag[i] ? 0 : 1].push_back(i);
          for (int k = 0; k < 2; k++) ds->d_blist[k] = dev_upload(lists[k].data(), lists[k].size()), ds->n_blist[k] = (int)lists[k].size();
        }
        if (ds->n_affine == nb || ds->d_blist[0]) {
          std::vector<double> wrel((size_t)Q4, 0.0);
          for (int i = 0; i < Q; i++) wrel[i] = geom->wq[i] / geom->wq[0];
          ds->d_affine = dev_upload(flag.data(), flag.size());
          ds->d_wrel = dev_upload(wrel.data(), wrel.size());
        }
      }
    }
  }
  PA_REQUIRE((dim == 3 && mode < MODE_CURL2 && !contra) || ds->d_L,
             "2-D blocks, div-div and div-div + mass need symmetric coefficients and tables that fit in LDS (fast path only)");
  return ds;
}

void free_dense_sub(DenseSub *ds) {
  if (ds) hipFree(ds->d_affine), hipFree(ds->d_wrel), hipFree(ds->d_ye2), hipFree(ds->d_blist[0]), hipFree(ds->d_blist[1]);
  if (!ds) return;
  hipFree(ds->d_idx), hipFree(ds->d_idx_bc), hipFree(ds->d_co), hipFree(ds->d_co2), hipFree(ds->d_ess_flag), hipFree(ds->d_rows);
  hipFree(ds->d_Tf), hipFree(ds->d_Tt), hipFree(ds->d_interp), hipFree(ds->d_deriv);
  hipFree(ds->d_L), hipFree(ds->d_qdata);
  hipFree(ds->d_off), hipFree(ds->d_cor), hipFree(ds->d_ori);
  hipFree(ds->d_ye), hipFree(ds->d_tptr), hipFree(ds->d_tent);
  hipFree(ds->d_rchunk), hipFree(ds->d_rhdr), hipFree(ds->d_rpos_run);
  hipFree(ds->c0.d_attr_mat), hipFree(ds->c0.d_mat), hipFree(ds->c0.d_mat_t);
  hipFree(ds->c1.d_attr_mat), hipFree(ds->c1.d_mat), hipFree(ds->c1.d_mat_t);
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
  std::vector<uint8_t> ef((size_t)ds.lsize, 0);
  for (int d = 0; d < ds.lsize && d < (int)flag.size(); d++) ef[d] = flag[d] ? 1 : 0;
  hipFree(ds.d_ess_flag);
  ds.d_ess_flag = dev_upload(ef.data(), ef.size());
}

// split-vector applies (pa_op_mult_split): the LDS-resident kernel and the staged one read x through dense_xbase
bool dense_split_ok(const DenseSub &ds) {
  return ds.d_ye && ds.d_tptr && ds.d_L && ds.PT <= 3 && !ds.d_blist[0] && resident_split_mode(ds.mode);
}

void launch_dense_apply(const DenseSub &ds, const double *x, bool masked, hipStream_t s, const SplitIO *split) {
  DenseArgs a = make_args(ds);
  a.x = x;
  if (split) {
    PA_REQUIRE(split->n_true >= 0 && split->n_true <= ds.lsize, "split point outside the local vector");
    a.nsplit = split->n_true;
    a.xg0 = split->xg0 - split->n_true, a.xg1 = (split->xg1 ? split->xg1 : split->xg0) - split->n_true;
    a.xg_sel = split->sel;
    PA_REQUIRE(dense_split_ok(ds), "no split-vector form of this dense block");
    if (masked && ds.d_idx_bc) a.idx = ds.d_idx_bc;
    switch (ds.PT) {
      case 1: launch_resident_split<1>(ds, a, s); break;
      case 2: launch_resident_split<2>(ds, a, s); break;
      default: launch_resident_split<3>(ds, a, s); break;
    }
    return;
  }
  if (masked && ds.d_idx_bc) a.idx = ds.d_idx_bc;
  if (ds.d_L) {
    switch (ds.PT) {
      case 1: launch_resident_pt<1>(ds, a, s); break;
      case 2: launch_resident_pt<2>(ds, a, s); break;
      case 3: launch_resident_pt<3>(ds, a, s); break;
      case 4: launch_resident_pt<4>(ds, a, s); break;
      case 6: launch_resident_pt<6>(ds, a, s); break;
      default: throw Error("resident dense kernel not instantiated for this element size");
    }
    PA_HIP(hipGetLastError());
    return;
  }
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

// ---- complex form (SURVEY.md 8(f)-1 on the non-tensor path) ---------------------------------------------------------------
// dr: curl-curl + mass in the resident form (the kernel's tables, index arrays, E-vector); di: mass, curl-curl or both on the same
// space and geometry; both affine (only the D of the first point of di is read) or both curved (round 4: its D at every point)
bool dense_complex_ok(const DenseSub &dr, const DenseSub &di) {
  static const bool enabled = !(getenv("PALACE_AMD_COMPLEX_FUSED") && atoi(getenv("PALACE_AMD_COMPLEX_FUSED")) == 0);
  if (!enabled || dr.fe_type != PA_FE_HCURL || di.fe_type != PA_FE_HCURL || dr.geom != di.geom || dr.geom->dim != 3) return false;
  // all blocks affine, or none, or (round 5) a mesh with both kinds whose two operators split their blocks the same way: then the
  // affine form runs on the list of the affine blocks and the curved form on the list of the others, like the real apply
  if (dr.mode != MODE_CURLMASS || !dr.d_L || dr.PT > 3) return false;
  if (!(di.mode == MODE_CURLMASS || di.mode == MODE_VMASS || di.mode == MODE_CURL) || !di.d_qdata) return false;
  if ((dr.d_affine != nullptr) != (di.d_affine != nullptr)) return false;
  if ((dr.d_blist[0] != nullptr) != (di.d_blist[0] != nullptr)) return false;
  if (dr.d_blist[0]) {
    if (dr.nb != di.nb || dr.n_blist[0] != di.n_blist[0] || dr.n_blist[1] != di.n_blist[1] || !dr.d_affine) return false;
    std::vector<uint8_t> fr((size_t)dr.nb), fi((size_t)di.nb);
    PA_HIP(hipMemcpy(fr.data(), dr.d_affine, fr.size(), hipMemcpyDeviceToHost));
    PA_HIP(hipMemcpy(fi.data(), di.d_affine, fi.size(), hipMemcpyDeviceToHost));
    if (fr != fi) return false;
  }
  if (dr.ne != di.ne || dr.P != di.P || dr.Q != di.Q || dr.lsize != di.lsize) return false;
  if (di.mode != MODE_CURL && di.chk_interp != dr.chk_interp) return false;
  if (di.mode != MODE_VMASS && di.chk_deriv != dr.chk_deriv) return false;
  return dr.h_idx == di.h_idx && dr.h_co == di.h_co;  // same restriction (host compare; the callers cache the answer)
}

template <int PT>
static void launch_resident_complex_pt(const DenseSub &dr, const DenseArgs &a, hipStream_t s) {
  const int rows = dr.L_rows;
  const size_t shm = sizeof(double) * ((size_t)(a.Q4 + 31) / 32 * 32 + (size_t)rows * ResidentStride<PT>::S +
                                      (dr.d_co ? (size_t)kResWaves * 4 * PT * 64 : 0));
  if (dr.d_blist[0] && !a.blist) {  // affine and curved blocks: one launch each on its list (they write disjoint E-vector blocks)
    DenseArgs aa = a, ag = a;
    aa.blist = dr.d_blist[0], aa.nblist = dr.n_blist[0];
    ag.blist = dr.d_blist[1], ag.nblist = dr.n_blist[1], ag.affine = nullptr;
    launch_resident_complex_pt<PT>(dr, aa, s);
    launch_resident_complex_pt<PT>(dr, ag, s);
    return;
  }
  const int nblocks = a.blist ? a.nblist : dr.nb;
  if (nblocks == 0) return;
  int grid = (2 * nblocks + kResWaves - 1) / kResWaves;
  if (grid > dr.num_cu) grid = dr.num_cu;
  static std::atomic<bool> attr_set{false};  // idempotent set-up, may race between rank threads
  if (!attr_set) {
    PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE_CURLMASS, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE_CURLMASS, false, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE_CURLMASS, true, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PA_HIP(hipFuncSetAttribute((const void *)dense_apply_resident_kernel<PT, MODE_CURLMASS, false, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if (a.blist && a.affine)
    hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE_CURLMASS, true, true, true>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows);
  else if (a.blist)
    hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE_CURLMASS, false, true, true>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows);
  else if (a.affine)
    hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE_CURLMASS, true, true>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows);
  else
    hipLaunchKernelGGL((dense_apply_resident_kernel<PT, MODE_CURLMASS, false, true>), dim3(grid), dim3(64 * kResWaves), shm, s, a, rows);
}

void launch_dense_complex(const DenseSub &dr, const DenseSub &di, const double *xr, const double *xi, double *ye_i, hipStream_t s,
                          bool masked) {
  DenseArgs a = make_args(dr);
  a.x = xr, a.x1 = xi, a.ye1 = ye_i;
  if (masked && dr.d_idx_bc) a.idx = dr.d_idx_bc;  // essential entries of both parts read as zero
  a.qdata_i = di.d_qdata, a.ncq_i = di.ncq;
  a.qi_mass = di.mode == MODE_CURL ? -1 : 0;
  a.qi_curl = di.mode == MODE_CURL ? 0 : (di.mode == MODE_CURLMASS ? 6 : -1);
  switch (dr.PT) {
    case 1: launch_resident_complex_pt<1>(dr, a, s); break;
    case 2: launch_resident_complex_pt<2>(dr, a, s); break;
    case 3: launch_resident_complex_pt<3>(dr, a, s); break;
    default: throw Error("complex dense kernel not instantiated for this element size");
  }
  PA_HIP(hipGetLastError());
}

// E^T of a split-vector apply: the sum of et_gather_kernel (same order, hence the same bits), rows >= nsplit to the ghost-row
// buffer, ParOperator's essential rows fixed on the way (rap.cpp:223-233)
// STEP (round 6): the sum is consumed by a smoother step / residual (GatherStep, pa_internal.hpp) instead of being stored -- the dense
// gather owns every row, so the epilogue is all there is to it; interface dofs of a multi-rank apply leave their partial sum in
// t_iface for the halo kernel, ghost rows go to yg as in the plain form
// DOFS dofs per thread, a block width apart, and the copies of each in chunks of four: index words, then E-vector entries, then the
// vectors of the step side by side (one dof per thread walked its copies one dependent load pair at a time: latency-bound; the
// transfer gathers of pa_interp.hip got 2x from the same change).  A dof's copies are added in their order, an absent copy adds
// an exact zero: the sums are those of the one-dof form.
[[maybe_unused]] constexpr int kGatherWideFrom = 1 << 30;  // (DOFS = 4 is SLOWER on tetrahedra -- 46 -> 60 us, the longest of 4 x 64 rows sets a wave's
                                         // trip count -- one dof per thread with its copies in chunks of four: 46 -> 40 us)
template <bool STEP, int DOFS = 1>
__global__ __launch_bounds__(256) void et_gather_split_kernel_t(const int n, const int32_t *__restrict__ tptr,
                                                                const int32_t *__restrict__ tent, const double *__restrict__ ye,
                                                                double *__restrict__ y, double *__restrict__ yg, const int nsplit,
                                                                const uint8_t *__restrict__ ess, const double *__restrict__ x,
                                                                const int ess_policy, const GatherStep st) {
  const int d0 = blockIdx.x * (256 * DOFS) + threadIdx.x;
  int b[DOFS], e[DOFS];
  double s[DOFS];
  bool fixed[DOFS];  // ParOperator's essential row (rap.cpp:223-233): x or 0, no copies to add
  int longest = 0;
#pragma unroll
  for (int u = 0; u < DOFS; u++) {
    const int d = d0 + 256 * u;
    b[u] = d < n ? tptr[d] : 0, e[u] = d < n ? tptr[d + 1] : 0;
    fixed[u] = d < n && ess_policy >= 0 && ess && ess[d] && d < nsplit;
    s[u] = 0.0;
  }
#pragma unroll
  for (int u = 0; u < DOFS; u++) {
    if (fixed[u]) {
      s[u] = ess_policy ? x[d0 + 256 * u] : 0.0;
      e[u] = b[u];
    }
    longest = max(longest, e[u] - b[u]);
  }
  for (int q0 = 0; q0 < longest; q0 += 4) {
    int t[DOFS][4];
    double v[DOFS][4];
#pragma unroll
    for (int u = 0; u < DOFS; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) t[u][q] = b[u] + q0 + q < e[u] ? tent[b[u] + q0 + q] : 0;
#pragma unroll
    for (int u = 0; u < DOFS; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int tt = t[u][q];
        const double w = b[u] + q0 + q < e[u] ? ye[tt >= 0 ? tt : -1 - tt] : 0.0;
        v[u][q] = tt >= 0 ? w : -w;
      }
#pragma unroll
    for (int u = 0; u < DOFS; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) s[u] += v[u][q];
  }
  if (!STEP) {
#pragma unroll
    for (int u = 0; u < DOFS; u++) {
      const int d = d0 + 256 * u;
      if (d < n) (d < nsplit ? y : yg)[d] = s[u];
    }
    return;
  }
  // the step: 0 = ghost row (to yg), 1 = interface dof (partial sum to t_iface), 2 = consumed here
  int kind[DOFS];
  double r0[DOFS], dinv[DOFS], ev[DOFS], epv[DOFS], old[DOFS];
#pragma unroll
  for (int u = 0; u < DOFS; u++) {
    const int d = d0 + 256 * u;
    kind[u] = d >= n ? -1 : d >= nsplit ? 0 : (st.iface_mask && (st.iface_mask[d] & 2)) ? 1 : 2;
  }
#pragma unroll
  for (int u = 0; u < DOFS; u++) {
    const int d = d0 + 256 * u;
    const bool use = kind[u] == 2;
    r0[u] = use ? st.r0[d] : 0.0;
    dinv[u] = (use && (st.mode != 2 || st.out)) ? st.dinv[d] : 0.0;
    ev[u] = (use && st.mode != 2) ? x[d] : 0.0;
    epv[u] = (use && st.mode != 2 && st.ep) ? st.ep[d] : 0.0;
    old[u] = (use && st.mode != 2 && st.add) ? st.out[d] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < DOFS; u++) {
    const int d = d0 + 256 * u;
    if (kind[u] == 0) {
      yg[d] = s[u];
    } else if (kind[u] == 1) {
      st.t_iface[d] = s[u];
    } else if (kind[u] == 2) {
      if (st.t_add && !fixed[u]) s[u] += st.t_add[d];  // (the surface blocks' contributions)
      const double rv = r0[u] - s[u];
      if (st.mode == 2) {
        if (st.res) st.res[d] = rv;
        if (st.out) st.out[d] = st.sr * dinv[u] * rv;
      } else {
        double dk = st.sr * dinv[u] * rv;
        dk += st.sd * (ev[u] - epv[u]);
        st.out[d] = old[u] + (ev[u] + dk);
      }
    }
  }
}

// The same gather with G lanes per dof (G = 2, 4, 8: a power of two near the average number of copies).  On tetrahedra the
// number of copies varies widely -- an order-2 H1 vertex dof has ~24, its edge dofs ~5, and a wave of the one-thread-per-dof form
// runs as long as its longest row (159k dofs took 26 us) -- so the copies of a dof are dealt to the lanes of its group (lane g
// adds copies g, g + G, ... in order), the group's partial sums are combined by a butterfly, and lane 0 of the group does the
// epilogue.  Deterministic (fixed assignment, fixed combination order), but NOT the bits of the serial order.
template <bool STEP, int G>
__global__ __launch_bounds__(256) void et_gather_group_kernel(const int n, const int32_t *__restrict__ tptr,
                                                              const int32_t *__restrict__ tent, const double *__restrict__ ye,
                                                              double *__restrict__ y, double *__restrict__ yg, const int nsplit,
                                                              const uint8_t *__restrict__ ess, const double *__restrict__ x,
                                                              const int ess_policy, const GatherStep st) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int d = (int)(tid / G), g = (int)(tid % G);
  const bool live = d < n;
  int b = live ? tptr[d] : 0, e = live ? tptr[d + 1] : 0;
  const bool fixed = live && ess_policy >= 0 && ess && ess[d] && d < nsplit;  // ParOperator's essential row (rap.cpp:223-233)
  if (fixed) e = b;
  double s = 0.0;
  for (int k = b + g; k < e; k += 2 * G) {  // two of the lane's copies side by side
    const int t0 = tent[k], t1 = k + G < e ? tent[k + G] : 0;
    const double v0 = ye[t0 >= 0 ? t0 : -1 - t0];
    const double v1 = k + G < e ? ye[t1 >= 0 ? t1 : -1 - t1] : 0.0;
    s += t0 >= 0 ? v0 : -v0;
    s += t1 >= 0 ? v1 : -v1;
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (!live || g != 0) return;
  if (fixed) s = ess_policy ? x[d] : 0.0;
  if (!STEP) {
    (d < nsplit ? y : yg)[d] = s;
    return;
  }
  if (d >= nsplit) {
    yg[d] = s;
    return;
  }
  if (st.iface_mask && (st.iface_mask[d] & 2)) {
    st.t_iface[d] = s;
    return;
  }
  if (st.t_add && !fixed) s += st.t_add[d];  // (the surface blocks' contributions)
  const double rv = st.r0[d] - s;
  if (st.mode == 2) {
    if (st.res) st.res[d] = rv;
    if (st.out) st.out[d] = st.sr * st.dinv[d] * rv;
    return;
  }
  const double ev = x[d];
  double dk = st.sr * st.dinv[d] * rv;
  dk += st.sd * (ev - (st.ep ? st.ep[d] : 0.0));
  st.out[d] = (st.add ? st.out[d] : 0.0) + (ev + dk);
}
// lanes per dof for a block with `nnz` E-vector entries on `n` dofs (PALACE_AMD_DENSE_GATHER_GROUP=0: the one-thread-per-dof form)
int gather_group(long long nnz, int n) {
  const char *e = std::getenv("PALACE_AMD_DENSE_GATHER_GROUP");
  if (e && e[0] == '0') return 1;
  if (e && (e[0] == '2' || e[0] == '4' || e[0] == '8') && !e[1]) return e[0] - '0';
  // (a deterministic rule, not a timing: the group size changes the order of the sums.  Measured: order-3 Nedelec tetrahedra, 2.4
  // copies per dof, gain nothing from two lanes per dof on either mesh -- 62 -> 64 us on config 3's, 0.184 -> 0.195 ms per apply on the
  // Kuhn cube; spaces with vertex dofs, >= 3 copies on average, gain up to 2x)
  const double avg = n > 0 ? (double)nnz / n : 1.0;
  return avg < 3.0 ? 1 : avg < 5.5 ? 4 : 8;
}
int dense_gather_group(const DenseSub &ds) { return gather_group((long long)ds.ne * ds.P, ds.lsize); }
template <bool STEP>
void launch_gather_split(const DenseSub &ds, const double *ye, double *y, double *yg, int nsplit, const double *x, int ess_policy,
                         const GatherStep &st, hipStream_t s) {
  const int G = gather_group((long long)ds.ne * ds.P, ds.lsize);
  const unsigned nb = (unsigned)(((long long)ds.lsize * G + 255) / 256);
#define PA_GATHER_GROUP(GG)                                                                                                         \
  hipLaunchKernelGGL((et_gather_group_kernel<STEP, GG>), dim3(nb), dim3(256), 0, s, ds.lsize, ds.d_tptr, ds.d_tent, ye, y, yg, nsplit, \
                     ds.d_ess_flag, x, ess_policy, st)
  if (G == 8) PA_GATHER_GROUP(8);
  else if (G == 4) PA_GATHER_GROUP(4);
  else if (G == 2) PA_GATHER_GROUP(2);
  else
    hipLaunchKernelGGL((et_gather_split_kernel_t<STEP, 1>), dim3(nb), dim3(256), 0, s, ds.lsize, ds.d_tptr, ds.d_tent, ye, y, yg, nsplit,
                       ds.d_ess_flag, x, ess_policy, st);
#undef PA_GATHER_GROUP
}

double time_dense_gather(const DenseSub &ds, const int32_t *d_tent) {
  DenseSub probe;  // (only what launch_gather_split reads: sizes and the map under test; nothing of it is freed here)
  probe.ne = ds.ne, probe.P = ds.P, probe.lsize = ds.lsize, probe.d_tptr = ds.d_tptr;
  probe.d_tent = const_cast<int32_t *>(d_tent);
  double *y = dev_alloc<double>((size_t)ds.lsize);
  hipEvent_t e0, e1;
  PA_HIP(hipEventCreate(&e0));
  PA_HIP(hipEventCreate(&e1));
  const int reps = 5;
  for (int it = 0; it < 2; it++)
    launch_gather_split<false>(probe, ds.d_ye, y, y, 0x7fffffff, nullptr, -1, GatherStep{}, nullptr);
  PA_HIP(hipEventRecord(e0, nullptr));
  for (int it = 0; it < reps; it++)
    launch_gather_split<false>(probe, ds.d_ye, y, y, 0x7fffffff, nullptr, -1, GatherStep{}, nullptr);
  PA_HIP(hipEventRecord(e1, nullptr));
  PA_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PA_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
  (void)hipFree(y);
  return ms / reps;
}

// E^T of the dense path by runs (pa_stream_host.hpp: build_runs_dense): one thread per L-dof, its run and offset from the chunk
// masks (12 bytes per 64 dofs), ONE position word per run and copy instead of one per dof and copy -- the copies of the dofs of an
// edge or a face sit 16 doubles apart in the element's E-vector column, forwards or backwards -- same copies in the same order as
// the CSR form (et_gather_kernel), hence the same bits.  kDenseGatherILP dofs per thread, a block width apart, their loads side by
// side.  Split vectors (rows >= nsplit to yg) and ParOperator's essential rows (ess flags + policy) as in et_gather_split_kernel.
constexpr int kDenseGatherILP = 4;
__global__ __launch_bounds__(256) void et_run_gather_dense_kernel(const int n, const streamhost::RunChunk *__restrict__ chunk,
                                                                  const streamhost::RunHdr *__restrict__ hdr,
                                                                  const uint32_t *__restrict__ rpos, const double *__restrict__ ye,
                                                                  double *__restrict__ y, double *__restrict__ yg, const int nsplit,
                                                                  const int accumulate, const uint8_t *__restrict__ ess,
                                                                  const double *__restrict__ x, const int ess_policy) {
  const int k0 = blockIdx.x * (256 * kDenseGatherILP) + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int pb[kDenseGatherILP], pe[kDenseGatherILP], j[kDenseGatherILP], run[kDenseGatherILP];
  bool live[kDenseGatherILP];
  double s[kDenseGatherILP], yold[kDenseGatherILP];
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++) {
    const int k = k0 + 256 * u;
    live[u] = k < n;
    const streamhost::RunChunk c = chunk[live[u] ? (k >> 6) : 0];
    const unsigned long long low = (c.starts & ~1ull) & ((2ull << lane) - 1ull);
    const int nc = __popcll(low);
    run[u] = (int)(c.first >> 4) + nc;
    j[u] = nc ? lane - (63 - __clzll((long long)low)) : (int)(c.first & 15u) + lane;
  }
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++) {
    pb[u] = hdr[live[u] ? run[u] : 0].ptr;
    pe[u] = hdr[live[u] ? run[u] + 1 : 0].ptr;
    s[u] = 0.0, yold[u] = 0.0;
    if (live[u] && ess_policy >= 0 && ess && ess[k0 + 256 * u] && k0 + 256 * u < nsplit) {
      s[u] = ess_policy ? x[k0 + 256 * u] : 0.0;  // essential row: no copies to sum
      pe[u] = pb[u];
    } else if (live[u] && accumulate) {
      yold[u] = y[k0 + 256 * u];
    }
  }
  unsigned r4[kDenseGatherILP][4];
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++)
#pragma unroll
    for (int q = 0; q < 4; q++) r4[u][q] = (live[u] && pb[u] + q < pe[u]) ? rpos[pb[u] + q] : 0xffffffffu;
  double v[kDenseGatherILP][4];
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned r = r4[u][q];
      const bool have = live[u] && pb[u] + q < pe[u];
      const long long at = (long long)(r & streamhost::kDenseRunPosMask) + ((r & streamhost::kDenseRunBack) ? -16 * j[u] : 16 * j[u]);
      const double w = have ? ye[at] : 0.0;
      v[u][q] = (r & streamhost::kDenseRunNeg) ? -w : w;
    }
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++) {
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (live[u] && pb[u] + q < pe[u]) s[u] += v[u][q];
    if (live[u])
      for (int p = pb[u] + 4; p < pe[u]; p++) {
        const unsigned r = rpos[p];
        const double w = ye[(long long)(r & streamhost::kDenseRunPosMask) + ((r & streamhost::kDenseRunBack) ? -16 * j[u] : 16 * j[u])];
        s[u] += (r & streamhost::kDenseRunNeg) ? -w : w;
      }
  }
#pragma unroll
  for (int u = 0; u < kDenseGatherILP; u++) {
    const int d = k0 + 256 * u;
    if (live[u]) (d < nsplit ? y : yg)[d] = yold[u] + s[u];  // (y + sum of the copies: the CSR form's order)
  }
}

// y[d] += sign * sum of the copies of d (CSR form, the copies in element order), rows without copies and -- skip_ess -- essential rows
// untouched: the small sub-operators added to a vector another operator has already written (pa_op_mult_complex)
__global__ void et_gather_signed_kernel(const int n, const int32_t *__restrict__ rows, const int32_t *__restrict__ tptr,
                                        const int32_t *__restrict__ tent, const double *__restrict__ ye, double *__restrict__ y,
                                        const double sign, const uint8_t *__restrict__ ess) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int d = rows ? rows[i] : i;  // (a block with few rows walks its row list)
  const int k0 = tptr[d], k1 = tptr[d + 1];
  if (k0 == k1 || (ess && ess[d])) return;
  double s = 0.0;
  for (int k = k0; k < k1; k++) {
    const int t = tent[k];
    const double v = ye[t >= 0 ? t : -1 - t];
    s += t >= 0 ? v : -v;
  }
  y[d] += sign * s;
}
void launch_dense_gather_signed(const DenseSub &ds, double *y, double sign, bool skip_ess, hipStream_t s) {
  if (ds.lsize == 0) return;
  PA_REQUIRE(!skip_ess || ds.d_ess_flag, "essential rows: pa_op_set_essential first");
  const int nr = ds.d_rows ? ds.n_rows : ds.lsize;
  if (nr == 0) return;
  hipLaunchKernelGGL(et_gather_signed_kernel, dim3((nr + 255) / 256), dim3(256), 0, s, nr, ds.d_rows, ds.d_tptr, ds.d_tent, ds.d_ye, y,
                     sign, skip_ess ? ds.d_ess_flag : nullptr);
  PA_HIP(hipGetLastError());
}

void launch_dense_gather(const DenseSub &ds, double *y, bool accumulate, hipStream_t s, const double *ye, const SplitIO *split,
                         const double *x, int ess_policy) {
  if (ds.d_rchunk) {
    PA_REQUIRE(!(split || ess_policy >= 0) || !accumulate, "split vectors / fused essential rows: y = A x only");
    PA_REQUIRE(ess_policy < 0 || (ds.d_ess_flag && x), "essential rows fused into the gather: pa_op_set_essential first");
    const int n = ds.lsize;
    if (n == 0) return;
    hipLaunchKernelGGL(et_run_gather_dense_kernel, dim3((n + 256 * kDenseGatherILP - 1) / (256 * kDenseGatherILP)), dim3(256), 0, s, n,
                       reinterpret_cast<const streamhost::RunChunk *>(ds.d_rchunk),
                       reinterpret_cast<const streamhost::RunHdr *>(ds.d_rhdr), ds.d_rpos_run, ye ? ye : ds.d_ye, y,
                       split ? split->yg - split->n_true : y, split ? split->n_true : 0x7fffffff, accumulate ? 1 : 0,
                       ess_policy >= 0 ? ds.d_ess_flag : nullptr, x, ess_policy);
    PA_HIP(hipGetLastError());
    return;
  }
  // every overwriting gather takes the same kernel (plain, split vectors, essential rows fixed on the way): y = A x has the same bits
  // whichever entry point computed it (tests/test_split_gpu.py compares them with torch.equal); the accumulating forms below keep the
  // serial order of et_gather_kernel
  if (split || ess_policy >= 0 || !accumulate) {
    PA_REQUIRE(!accumulate, "split vectors / fused essential rows: y = A x only");
    PA_REQUIRE(ess_policy < 0 || (ds.d_ess_flag && x), "essential rows fused into the gather: pa_op_set_essential first");
    launch_gather_split<false>(ds, ye ? ye : ds.d_ye, y, split ? split->yg - split->n_true : y, split ? split->n_true : 0x7fffffff, x,
                               ess_policy, GatherStep{}, s);
    PA_HIP(hipGetLastError());
    return;
  }
  if (ds.d_rows && accumulate) {  // (few rows: only those; an overwriting apply still takes the pass over the whole vector)
    launch_et_gather_raw(ds.n_rows, ds.d_tptr, ds.d_tent, ye ? ye : ds.d_ye, y, true, s, ds.d_rows);
    return;
  }
  if (ye) {
    launch_et_gather_raw(ds.lsize, ds.d_tptr, ds.d_tent, ye, y, accumulate, s);
    return;
  }
  launch_et_gather_raw(ds.lsize, ds.d_tptr, ds.d_tent, ds.d_ye, y, accumulate, s);
}

__global__ void k_zero_rows(double *__restrict__ v, const int32_t *__restrict__ rows, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[rows[i]] = 0.0;
}
void launch_zero_rows(double *v, const int32_t *rows, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_zero_rows, dim3((n + 255) / 256), dim3(256), 0, s, v, rows, n);
  PA_HIP(hipGetLastError());
}

__global__ void k_surface_rows(const int32_t *__restrict__ rows, const int n, const int32_t *__restrict__ row_ptr,
                               const int32_t *__restrict__ ent, const uint8_t *__restrict__ blk, const SurfaceYe ye,
                               double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
    const int t = ent[e];
    const double v = ye.ye[blk[e]][t >= 0 ? t : -1 - t];
    s += t >= 0 ? v : -v;
  }
  out[rows[i]] = s;
}
void launch_surface_rows(const int32_t *rows, int n, const int32_t *row_ptr, const int32_t *ent, const uint8_t *blk, const SurfaceYe &ye,
                         double *out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_surface_rows, dim3((n + 255) / 256), dim3(256), 0, s, rows, n, row_ptr, ent, blk, ye, out);
  PA_HIP(hipGetLastError());
}

bool dense_fused_step_ok(const DenseSub &ds) { return ds.d_ess_flag && !ds.d_rchunk && ds.d_tptr && ds.d_tent && ds.d_ye; }
void launch_dense_gather_step(const DenseSub &ds, const double *x, const GatherStep &step, int ess_policy, hipStream_t s,
                              const SplitIO *split) {
  PA_REQUIRE(dense_fused_step_ok(ds) && x, "dense fused step: essential list fused (pa_op_set_essential) and the CSR-form gather expected");
  PA_REQUIRE(!split || (step.iface_mask && step.t_iface), "split form of the fused step: interface mask and buffer missing");
  if (ds.lsize == 0) return;
  launch_gather_split<true>(ds, ds.d_ye, nullptr, split ? split->yg - split->n_true : nullptr, split ? split->n_true : 0x7fffffff, x,
                            ess_policy, step, s);
  PA_HIP(hipGetLastError());
}

void launch_dense_diag(const DenseSub &ds, double *diag_out, hipStream_t s) {
  DenseArgs a = make_args(ds);
  const long long n = (long long)ds.ne * ds.P;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  double *diag = dev_alloc<double>((size_t)n);  // element diagonals [ne][P] (set-up path: allocated per call)
  if (ds.geom->dim == 3 && ds.mode < MODE_CURL2 && !ds.contra) switch (ds.mode) {
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
    default: break;
  }
  else switch (ds.mode) {  // 2-D blocks (and div-div) have q-data only
#define PA_DIAG2_CASE(MODE)                                                                                     \
  case MODE:                                                                                                    \
    hipLaunchKernelGGL((dense_diag_qd_kernel<MODE>), grid, block, 0, s, a, ds.d_off, ds.d_cor, ds.d_interp, ds.d_deriv, \
                       diag);                                                                                   \
    break;
    PA_DIAG2_CASE(MODE_CURL2)
    PA_DIAG2_CASE(MODE_VMASS2)
    PA_DIAG2_CASE(MODE_CURLMASS2)
    PA_DIAG2_CASE(MODE_MASS)
    PA_DIAG2_CASE(MODE_DIFF2)
    PA_DIAG2_CASE(MODE_DIFFMASS2)
    PA_DIAG2_CASE(MODE_DIFFMASS)  // (l2mass_33: the div-div + mass pair on 3-D H(div) elements has q-data only)
    PA_DIAG2_CASE(MODE_VMASS1)
    PA_DIAG2_CASE(MODE_DIFF1)
    PA_DIAG2_CASE(MODE_DIFFMASS1)
#undef PA_DIAG2_CASE
    default: break;
  }
  hipLaunchKernelGGL(dense_diag_slot_kernel, grid, block, 0, s, ds.ne, ds.P, ds.KP, ds.d_cor, ds.d_idx, diag, ds.d_ye, ds.ye_rows ? 1 : 0);
  PA_HIP(hipGetLastError());
  launch_et_gather_raw(ds.lsize, ds.d_tptr, ds.d_tent, ds.d_ye, diag_out, true, s);
  PA_HIP(hipStreamSynchronize(s));
  PA_HIP(hipFree(diag));
}

}  // namespace pa
