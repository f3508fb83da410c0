// Fused E -> B/G -> D -> B^T/G^T for H1 tensor-product hexahedra (gfx950, FP64): diffusion, mass and
// diffusion+mass (reference integrators fem/integ/{diffusion,mass,diffusionmass}.cpp; D from
// fem/qfunctions/33/hcurl_33_qf.h applied to grad u, fem/qfunctions/1/h1_1_qf.h and
// fem/qfunctions/33/hcurlmass_33_qf.h).  Palace gives libCEED a tensor H1 basis for these elements
// (fem/libceed/basis.cpp:15-38) and the lexicographic restriction (restriction.cpp:113-205).
//
// Same mapping as pa_nd_hex.hip: Q1^2 lanes per element, passes X -> Y -> Z through LDS inside the
// wave, lane (qx,qy) ends with its qz column of u and grad u; E^T is the E-vector + gather form.
#include "pa_internal.hpp"

namespace pa {

template <int P1, int Q1>
struct H1Tab {
  static constexpr int QH = (Q1 + 1) / 2;  // mirror symmetry, see pa_nd_hex.hip
  double Bc[QH * (P1 + 1)];
  double Gc[QH * (P1 + 1)];
};

template <int N, int Q1>
__device__ __forceinline__ double h1_even(const double *H, const int q, const int i) {
  return (q < (Q1 + 1) / 2) ? H[q * N + i] : H[(Q1 - 1 - q) * N + (N - 1 - i)];
}
template <int N, int Q1>
__device__ __forceinline__ double h1_odd(const double *H, const int q, const int i) {
  return (q < (Q1 + 1) / 2) ? H[q * N + i] : -H[(Q1 - 1 - q) * N + (N - 1 - i)];
}

template <int P1, int Q1>
struct H1Args {
  int ne;
  const int32_t *sidx_in;  // sorted-order index (see pa_nd_hex.hip), kEssBit = read as zero
  const uint16_t *perm;    // tensor-order slot of sorted entry m
  const double *geom;
  const double *qdata;
  const double *x;
  double *ye;
  CoeffDev c_mass, c_diff;
  H1Tab<P1, Q1> tab;
};

__device__ __forceinline__ void h1_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// utils_33_qf.h:64-84: y = s A^T B C x
__device__ __forceinline__ void h1_AtBCx33(const double A[9], const double B[9], const double C[9], const double x0,
                                           const double x1, const double x2, const double s, double &y0, double &y1,
                                           double &y2) {
  const double t0 = C[0] * x0 + C[3] * x1 + C[6] * x2;
  const double t1 = C[1] * x0 + C[4] * x1 + C[7] * x2;
  const double t2 = C[2] * x0 + C[5] * x1 + C[8] * x2;
  const double z0 = B[0] * t0 + B[3] * t1 + B[6] * t2;
  const double z1 = B[1] * t0 + B[4] * t1 + B[7] * t2;
  const double z2 = B[2] * t0 + B[5] * t1 + B[8] * t2;
  y0 = s * (A[0] * z0 + A[1] * z1 + A[2] * z2);
  y1 = s * (A[3] * z0 + A[4] * z1 + A[5] * z2);
  y2 = s * (A[6] * z0 + A[7] * z1 + A[8] * z2);
}

template <int P1, int Q1>
struct H1Layout {
  static constexpr int NC = P1 + 1;
  static constexpr int T = Q1 * Q1;
  static constexpr int EPW = 64 / T;
  static constexpr int A_FIELD = Q1 * NC * NC;
  static constexpr int B_FIELD = Q1 * Q1 * NC;
  static constexpr int ELEM = 2 * A_FIELD + 3 * B_FIELD;
  static constexpr int ELEM_PAD = ((ELEM + 15) / 16 * 16) | 16;
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) { return f * A_FIELD + (qx * NC + j) * NC + k; }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) {
    return 2 * A_FIELD + f * B_FIELD + (qx * Q1 + qy) * NC + k;
  }
};

constexpr int kH1Waves = 4;

// USE_V: mass term (value), USE_G: diffusion term (gradient); QD: packed pre-assembled D
// (mass: 1 double c w detJ; diffusion: 6 doubles of w detJ adj^T C adj)
template <int P1, int Q1, bool USE_V, bool USE_G, bool QD>
__global__ __launch_bounds__(64 * kH1Waves, 2) void h1_hex_apply_kernel(const H1Args<P1, Q1> a) {
  using L = H1Layout<P1, Q1>;
  constexpr int NC = L::NC, Q = Q1 * Q1 * Q1, P = NC * NC * NC;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / L::T, t = lane - sub * L::T;
  const int ta = t % Q1, tb = t / Q1;
  const bool lane_ok = sub < L::EPW;
  const int e = (blockIdx.x * kH1Waves + wave) * L::EPW + sub;
  const bool active = lane_ok && e < a.ne;
  double *sm = smem + (size_t)(wave * L::EPW + (lane_ok ? sub : 0)) * L::ELEM_PAD;
  const double *Bc = a.tab.Bc, *Gc = a.tab.Gc;

  // geometry / q-data of this lane's Q1 points, requested up front
  constexpr int NG = QD ? (USE_V ? 1 : 0) + (USE_G ? 6 : 0) : 10;
  double gd[Q1][NG];
  int attr[Q1];
  if (QD) {
    const double *g = a.qdata + (size_t)(active ? e : 0) * NG * Q + ta + Q1 * tb;
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      attr[qz] = 0;
#pragma unroll
      for (int c = 0; c < NG; c++) gd[qz][c] = g[c * Q + Q1 * Q1 * qz];
    }
  } else {
    const double *g = a.geom + (size_t)(active ? e : 0) * 11 * Q + ta + Q1 * tb;
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      attr[qz] = (int)g[Q1 * Q1 * qz];
#pragma unroll
      for (int c = 0; c < 10; c++) gd[qz][c] = g[(1 + c) * Q + Q1 * Q1 * qz];
    }
  }

  // E: sorted-order gather staged through LDS into tensor order
  constexpr int NPL = (P + L::T - 1) / L::T;
  int lp[NPL];
#pragma unroll
  for (int r = 0; r < NPL; r++) {
    const int m = t + L::T * r;
    lp[r] = 0;
    if (active && m < P) {
      const int s = a.sidx_in[(size_t)e * P + m];
      lp[r] = a.perm[(size_t)e * P + m];
      sm[lp[r]] = (s & kEssBit) ? 0.0 : a.x[s & ~kEssBit];
    }
  }
  h1_wave_sync();
  double u[NC];
  {
    const bool act = ta < NC && tb < NC;
#pragma unroll
    for (int i = 0; i < NC; i++) u[i] = act ? sm[i + NC * (ta + NC * tb)] : 0.0;
  }
  h1_wave_sync();

  double V[Q1], GV[3][Q1];
  // ---- forward: pass X, lane (j, k)
  {
    const bool act = ta < NC && tb < NC;
#pragma unroll
    for (int qx = 0; qx < Q1; qx++) {
      double v = 0.0, d = 0.0;
#pragma unroll
      for (int i = 0; i < NC; i++) {
        v += h1_even<NC, Q1>(Bc, qx, i) * u[i];
        if (USE_G) d += h1_odd<NC, Q1>(Gc, qx, i) * u[i];
      }
      if (lane_ok && act) {
        sm[L::ia(0, qx, ta, tb)] = v;
        if (USE_G) sm[L::ia(1, qx, ta, tb)] = d;
      }
    }
  }
  h1_wave_sync();
  // pass Y, lane (qx, k)
  {
    const bool act = tb < NC;
    double v[NC], d[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
      v[j] = sm[L::ia(0, ta, j, act ? tb : 0)];
      if (USE_G) d[j] = sm[L::ia(1, ta, j, act ? tb : 0)];
    }
#pragma unroll
    for (int qy = 0; qy < Q1; qy++) {
      double vv = 0.0, vd = 0.0, dv = 0.0;
#pragma unroll
      for (int j = 0; j < NC; j++) {
        vv += h1_even<NC, Q1>(Bc, qy, j) * v[j];
        if (USE_G) vd += h1_odd<NC, Q1>(Gc, qy, j) * v[j];
        if (USE_G) dv += h1_even<NC, Q1>(Bc, qy, j) * d[j];
      }
      if (lane_ok && act) {
        sm[L::ib(0, ta, qy, tb)] = vv;
        if (USE_G) sm[L::ib(1, ta, qy, tb)] = vd, sm[L::ib(2, ta, qy, tb)] = dv;
      }
    }
  }
  h1_wave_sync();
  // pass Z, lane (qx, qy)
  {
    double vv[NC], vd[NC], dv[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) {
      vv[k] = sm[L::ib(0, ta, tb, k)];
      if (USE_G) vd[k] = sm[L::ib(1, ta, tb, k)], dv[k] = sm[L::ib(2, ta, tb, k)];
    }
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      double val = 0.0, dz = 0.0, dy = 0.0, dx = 0.0;
#pragma unroll
      for (int k = 0; k < NC; k++) {
        if (USE_V) val += h1_even<NC, Q1>(Bc, qz, k) * vv[k];
        if (USE_G) {
          dz += h1_odd<NC, Q1>(Gc, qz, k) * vv[k];
          dy += h1_even<NC, Q1>(Bc, qz, k) * vd[k];
          dx += h1_even<NC, Q1>(Bc, qz, k) * dv[k];
        }
      }
      V[qz] = val, GV[0][qz] = dx, GV[1][qz] = dy, GV[2][qz] = dz;
    }
  }
  h1_wave_sync();

  // ---- D (h1_1 / hcurl_33 on grad u / hcurlmass_33)
#pragma unroll
  for (int qz = 0; qz < Q1; qz++) {
    if (QD) {
      if (USE_V) V[qz] *= gd[qz][0];
      if (USE_G) {
        const double *m = &gd[qz][USE_V ? 1 : 0];
        const double x0 = GV[0][qz], x1 = GV[1][qz], x2 = GV[2][qz];
        GV[0][qz] = m[0] * x0 + m[1] * x1 + m[2] * x2;
        GV[1][qz] = m[1] * x0 + m[3] * x1 + m[4] * x2;
        GV[2][qz] = m[2] * x0 + m[4] * x1 + m[5] * x2;
      }
    } else {
      const double wdetJ = gd[qz][0];
      const double *adj = &gd[qz][1];
      if (USE_V) {
        const int k = (a.c_mass.nattr > 0) ? a.c_mass.attr_mat[attr[qz] - 1] : 0;
        V[qz] *= a.c_mass.mat[k] * wdetJ;  // CoeffUnpack1, coeff_1_qf.h
      }
      if (USE_G) {
        double Cm[9];
        const int k = (a.c_diff.nattr > 0) ? a.c_diff.attr_mat[attr[qz] - 1] : 0;
#pragma unroll
        for (int i = 0; i < 9; i++) Cm[i] = a.c_diff.mat[9 * k + i];
        h1_AtBCx33(adj, Cm, adj, GV[0][qz], GV[1][qz], GV[2][qz], wdetJ, GV[0][qz], GV[1][qz], GV[2][qz]);
      }
    }
  }

  // ---- transposed passes: Z^T lane (qx, qy)
  {
#pragma unroll
    for (int k = 0; k < NC; k++) {
      double vv = 0.0, vd = 0.0, dv = 0.0;
#pragma unroll
      for (int qz = 0; qz < Q1; qz++) {
        if (USE_V) vv += h1_even<NC, Q1>(Bc, qz, k) * V[qz];
        if (USE_G) {
          vv += h1_odd<NC, Q1>(Gc, qz, k) * GV[2][qz];
          vd += h1_even<NC, Q1>(Bc, qz, k) * GV[1][qz];
          dv += h1_even<NC, Q1>(Bc, qz, k) * GV[0][qz];
        }
      }
      if (lane_ok) {
        sm[L::ib(0, ta, tb, k)] = vv;
        if (USE_G) sm[L::ib(1, ta, tb, k)] = vd, sm[L::ib(2, ta, tb, k)] = dv;
      }
    }
  }
  h1_wave_sync();
  // Y^T lane (qx, k)
  {
    const bool act = tb < NC;
    double vv[Q1], vd[Q1], dv[Q1];
#pragma unroll
    for (int qy = 0; qy < Q1; qy++) {
      vv[qy] = sm[L::ib(0, ta, qy, act ? tb : 0)];
      if (USE_G) vd[qy] = sm[L::ib(1, ta, qy, act ? tb : 0)], dv[qy] = sm[L::ib(2, ta, qy, act ? tb : 0)];
    }
#pragma unroll
    for (int j = 0; j < NC; j++) {
      double v = 0.0, d = 0.0;
#pragma unroll
      for (int qy = 0; qy < Q1; qy++) {
        v += h1_even<NC, Q1>(Bc, qy, j) * vv[qy];
        if (USE_G) v += h1_odd<NC, Q1>(Gc, qy, j) * vd[qy];
        if (USE_G) d += h1_even<NC, Q1>(Bc, qy, j) * dv[qy];
      }
      if (lane_ok && act) {
        sm[L::ia(0, ta, j, tb)] = v;
        if (USE_G) sm[L::ia(1, ta, j, tb)] = d;
      }
    }
  }
  h1_wave_sync();
  // X^T lane (j, k) -> E-vector [i][j + NC k]
  {
    const bool act = ta < NC && tb < NC;
    double v[Q1], d[Q1];
#pragma unroll
    for (int qx = 0; qx < Q1; qx++) {
      v[qx] = sm[L::ia(0, qx, act ? ta : 0, act ? tb : 0)];
      if (USE_G) d[qx] = sm[L::ia(1, qx, act ? ta : 0, act ? tb : 0)];
    }
#pragma unroll
    for (int i = 0; i < NC; i++) {
      double r = 0.0;
#pragma unroll
      for (int qx = 0; qx < Q1; qx++) {
        r += h1_even<NC, Q1>(Bc, qx, i) * v[qx];
        if (USE_G) r += h1_odd<NC, Q1>(Gc, qx, i) * d[qx];
      }
      u[i] = r;
    }
  }
  h1_wave_sync();
  // E^T, first half: results into tensor order in LDS, then out in sorted order (coalesced)
  {
    const bool act = lane_ok && ta < NC && tb < NC;
#pragma unroll
    for (int i = 0; i < NC; i++)
      if (act) sm[i + NC * (ta + NC * tb)] = u[i];
  }
  h1_wave_sync();
#pragma unroll
  for (int r = 0; r < NPL; r++) {
    const int m = t + L::T * r;
    if (active && m < P) a.ye[(size_t)e * P + m] = sm[lp[r]];
  }
}

template <int P1, int Q1>
static void h1_launch_pq(const SubOp &so, const double *x, bool masked, hipStream_t s) {
  using L = H1Layout<P1, Q1>;
  constexpr int QH = H1Tab<P1, Q1>::QH;
  H1Args<P1, Q1> a;
  a.ne = so.ne;
  a.sidx_in = (masked && so.d_sidx_bc) ? so.d_sidx_bc : so.d_sidx;
  a.perm = so.d_perm;
  a.geom = so.geom->d_geom;
  a.qdata = so.qd ? so.qd->d : nullptr;
  a.x = x;
  a.ye = so.d_ye;
  for (int i = 0; i < QH * (P1 + 1); i++) a.tab.Bc[i] = so.Bc[i], a.tab.Gc[i] = so.Gc[i];
  const int epb = kH1Waves * L::EPW;
  const dim3 grid((so.ne + epb - 1) / epb), block(64 * kH1Waves);
  const size_t lds = sizeof(double) * (size_t)epb * L::ELEM_PAD;
  const bool qd = a.qdata != nullptr;
#define PA_H1_LAUNCH(V, G)                                                                    \
  if (qd)                                                                                     \
    hipLaunchKernelGGL((h1_hex_apply_kernel<P1, Q1, V, G, true>), grid, block, lds, s, a);    \
  else                                                                                        \
    hipLaunchKernelGGL((h1_hex_apply_kernel<P1, Q1, V, G, false>), grid, block, lds, s, a);
  switch (so.qf) {
    case PA_QF_HCURL_33:
      a.c_diff = so.c0.dev();
      PA_H1_LAUNCH(false, true)
      break;
    case PA_QF_H1_1:
      a.c_mass = so.c0.dev();
      PA_H1_LAUNCH(true, false)
      break;
    case PA_QF_HCURLMASS_33:
      a.c_mass = so.c0.dev();
      a.c_diff = so.c1.dev();
      PA_H1_LAUNCH(true, true)
      break;
    default:
      throw Error("QFunction not available for H1 hexahedra");
  }
#undef PA_H1_LAUNCH
  PA_HIP(hipGetLastError());
}

#define PA_H1_DISPATCH(FN, ...)                                                           \
  switch (so.p * 16 + so.q1d) {                                                            \
    case 1 * 16 + 2: FN<1, 2>(__VA_ARGS__); break;                                         \
    case 1 * 16 + 3: FN<1, 3>(__VA_ARGS__); break;                                         \
    case 2 * 16 + 3: FN<2, 3>(__VA_ARGS__); break;                                         \
    case 1 * 16 + 4: FN<1, 4>(__VA_ARGS__); break;                                         \
    case 2 * 16 + 4: FN<2, 4>(__VA_ARGS__); break;                                         \
    case 3 * 16 + 4: FN<3, 4>(__VA_ARGS__); break;                                         \
    case 1 * 16 + 5: FN<1, 5>(__VA_ARGS__); break;                                         \
    case 2 * 16 + 5: FN<2, 5>(__VA_ARGS__); break;                                         \
    case 3 * 16 + 5: FN<3, 5>(__VA_ARGS__); break;                                         \
    case 4 * 16 + 5: FN<4, 5>(__VA_ARGS__); break;                                         \
    default:                                                                               \
      throw Error("no H1 hex kernel for order " + std::to_string(so.p) + " with " +        \
                  std::to_string(so.q1d) + " points per direction");                       \
  }

// writes the E-vector so.d_ye; the caller follows with launch_et_gather
void launch_h1_hex_apply(const SubOp &so, const double *x, bool masked, hipStream_t s) {
  PA_REQUIRE(so.d_ye, "H1 blocks use the gather form of E^T");
  PA_H1_DISPATCH(h1_launch_pq, so, x, masked, s)
}

// ---- packed q-data and diagonal (set-up) ----------------------------------------------------------
__global__ void h1_hex_qdata_kernel(const int ne, const int Q, const double *__restrict__ geom, const CoeffDev c_mass,
                                    const CoeffDev c_diff, const int use_v, const int use_g, double *__restrict__ qd) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  const double *g = geom + (size_t)e * 11 * Q;
  const int ncomp = use_v + 6 * use_g;
  double *out = qd + (size_t)e * ncomp * Q + q;
  const int attr = (int)g[q];
  const double w = g[Q + q];
  int o = 0;
  if (use_v) {
    const int k = (c_mass.nattr > 0) ? c_mass.attr_mat[attr - 1] : 0;
    out[0] = c_mass.mat[k] * w;
    o = 1;
  }
  if (use_g) {
    double adj[9], Cm[9], M[9];
    for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + q];
    const int k = (c_diff.nattr > 0) ? c_diff.attr_mat[attr - 1] : 0;
    for (int i = 0; i < 9; i++) Cm[i] = c_diff.mat[9 * k + i];
    for (int col = 0; col < 3; col++)
      h1_AtBCx33(adj, Cm, adj, col == 0, col == 1, col == 2, w, M[0 + 3 * col], M[1 + 3 * col], M[2 + 3 * col]);
    out[(o + 0) * Q] = M[0];
    out[(o + 1) * Q] = 0.5 * (M[3] + M[1]);
    out[(o + 2) * Q] = 0.5 * (M[6] + M[2]);
    out[(o + 3) * Q] = M[4];
    out[(o + 4) * Q] = 0.5 * (M[7] + M[5]);
    out[(o + 5) * Q] = M[8];
  }
}

void launch_h1_hex_qdata(SubOp &so, hipStream_t s) {
  const bool use_v = so.qf == PA_QF_H1_1 || so.qf == PA_QF_HCURLMASS_33;
  const bool use_g = so.qf == PA_QF_HCURL_33 || so.qf == PA_QF_HCURLMASS_33;
  auto *qd = new QData;
  qd->ncomp = (int)use_v + 6 * (int)use_g;
  {  // padded to whole batches of four elements (the streaming kernel reads them), pad = 0
    const size_t nq = (size_t)((so.ne + 3) & ~3) * qd->ncomp * so.Q;
    qd->d = dev_alloc<double>(nq);
    PA_HIP(hipMemsetAsync(qd->d, 0, nq * sizeof(double), s));
  }
  CoeffDev cm{}, cd{};
  if (so.qf == PA_QF_H1_1) cm = so.c0.dev();
  if (so.qf == PA_QF_HCURL_33) cd = so.c0.dev();
  if (so.qf == PA_QF_HCURLMASS_33) cm = so.c0.dev(), cd = so.c1.dev();
  const long long n = (long long)so.ne * so.Q;
  hipLaunchKernelGGL(h1_hex_qdata_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, so.ne, so.Q,
                     so.geom->d_geom, cm, cd, (int)use_v, (int)use_g, qd->d);
  PA_HIP(hipGetLastError());
  so.qd = qd;
}

struct H1DiagArgs {
  int ne, p, q1;
  const int32_t *lidx;
  const uint16_t *perm;  // E-vector form (default): entry m of the element in sorted order, see nd_hex_diag_kernel
  double *ye;
  const double *geom;
  double *y;
  CoeffDev c_mass, c_diff;
  const double *Bc, *Gc;
  bool use_v, use_g;
};

__global__ void h1_hex_diag_kernel(const H1DiagArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int Q1 = a.q1, NC = a.p + 1, Q = Q1 * Q1 * Q1, P = NC * NC * NC;
  double *Mg = dsm, *Mv = dsm + 9 * Q;
  const int e = blockIdx.x;
  const double *g = a.geom + (size_t)e * 11 * Q;
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    double adj[9], Cm[9];
    const int attr = (int)g[q];
    const double w = g[Q + q];
    for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + q];
    double mv = 0.0;
    if (a.use_v) {
      const int k = (a.c_mass.nattr > 0) ? a.c_mass.attr_mat[attr - 1] : 0;
      mv = a.c_mass.mat[k] * w;
    }
    Mv[q] = mv;
    for (int col = 0; col < 3; col++) {
      double y0 = 0, y1 = 0, y2 = 0;
      if (a.use_g) {
        const int k = (a.c_diff.nattr > 0) ? a.c_diff.attr_mat[attr - 1] : 0;
        for (int i = 0; i < 9; i++) Cm[i] = a.c_diff.mat[9 * k + i];
        h1_AtBCx33(adj, Cm, adj, col == 0, col == 1, col == 2, w, y0, y1, y2);
      }
      Mg[9 * q + 0 + 3 * col] = y0, Mg[9 * q + 1 + 3 * col] = y1, Mg[9 * q + 2 + 3 * col] = y2;
    }
  }
  __syncthreads();
  for (int m = threadIdx.x; m < P; m += blockDim.x) {
    const int l = a.ye ? a.perm[(size_t)e * P + m] : m;
    const int i = l % NC, j = (l / NC) % NC, k = l / (NC * NC);
    double acc = 0.0;
    for (int qz = 0; qz < Q1; qz++)
      for (int qy = 0; qy < Q1; qy++)
        for (int qx = 0; qx < Q1; qx++) {
          const int q = qx + Q1 * (qy + Q1 * qz);
          const double bx = a.Bc[qx * NC + i], by = a.Bc[qy * NC + j], bz = a.Bc[qz * NC + k];
          const double gr[3] = {a.Gc[qx * NC + i] * by * bz, bx * a.Gc[qy * NC + j] * bz, bx * by * a.Gc[qz * NC + k]};
          const double f = bx * by * bz;
          acc += Mv[q] * f * f;
          for (int r2 = 0; r2 < 3; r2++)
            for (int c2 = 0; c2 < 3; c2++) acc += gr[r2] * Mg[9 * q + r2 + 3 * c2] * gr[c2];
        }
    if (a.ye)
      a.ye[(size_t)e * P + m] = acc;  // summed per dof by et_gather_kernel (fixed order)
    else
      atomicAdd(&a.y[a.lidx[(size_t)e * P + l]], acc);  // dofs of one element are distinct, different elements race
  }
}

void launch_h1_hex_diag(const SubOp &so, double *diag, hipStream_t s) {
  H1DiagArgs a;
  a.ne = so.ne, a.p = so.p, a.q1 = so.q1d;
  a.lidx = so.d_lidx;
  a.perm = so.d_perm, a.ye = (so.d_tptr && so.d_perm) ? so.d_ye : nullptr;
  a.geom = so.geom->d_geom;
  a.y = diag;
  const int nc = so.p + 1;
  a.Bc = so.d_tab + so.q1d * so.p, a.Gc = a.Bc + so.q1d * nc;
  a.use_v = a.use_g = false;
  switch (so.qf) {
    case PA_QF_HCURL_33: a.c_diff = so.c0.dev(), a.use_g = true; break;
    case PA_QF_H1_1: a.c_mass = so.c0.dev(), a.use_v = true; break;
    case PA_QF_HCURLMASS_33: a.c_mass = so.c0.dev(), a.c_diff = so.c1.dev(), a.use_v = a.use_g = true; break;
    default: throw Error("QFunction not available for H1 hexahedra");
  }
  const size_t lds = sizeof(double) * 10 * (size_t)so.Q;
  hipLaunchKernelGGL(h1_hex_diag_kernel, dim3(so.ne), dim3(128), lds, s, a);
  PA_HIP(hipGetLastError());
  if (a.ye) launch_et_gather_raw(so.lsize, so.d_tptr, so.d_tent, so.d_ye, diag, true, s);
}

}  // namespace pa
