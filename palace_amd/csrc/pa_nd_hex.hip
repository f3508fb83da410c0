// Fused E -> B/G -> D -> B^T/G^T -> E^T for Nedelec hexahedra (gfx950, FP64).
//
// Replaces what libCEED does inside CeedOperatorApplyAdd for Palace's curl-curl, ND-mass and
// curl-curl+mass integrators (reference fem/libceed/operator.cpp:148-178; integrators
// fem/integ/{curlcurl,vecfemass,curlcurlmass}.cpp; D from fem/qfunctions/33/{hdiv_33,hcurl_33,
// hdivmass_33}_qf.h).  The reference feeds libCEED dense [3Q x P] tables (fem/libceed/basis.cpp:
// 40-85), which makes p=3 compute-bound; here the tensor structure of the element is used
// (sum factorisation), the whole chain runs in one kernel and nothing but x, the index array and
// the geometry data is read from HBM, and only y is written (FP64 hardware atomics).
//
// Mapping (CDNA4, 64-lane waves): one element per Q1*Q1 lanes, 64/(Q1*Q1) elements per wave.
// Lane (a, b) owns one line of the element along the direction being contracted; the two
// re-distributions per component go through LDS and stay inside the wave, so there is no
// workgroup barrier anywhere: waves are independent and overlap each other's HBM latency.
//   pass X: lane (j,k)   contracts i  -> qx     tables of comp's x-direction
//   pass Y: lane (qx,k)  contracts j  -> qy
//   pass Z: lane (qx,qy) contracts k  -> qz     => lane (qx,qy) holds the qz column
// D runs on the qz column in registers; the transposed passes mirror the above and end in the
// signed scatter-add.  The geometry data of the lane's Q1 points (the dominant HBM stream) is
// requested first thing, so its latency hides behind the forward contraction.  The 1-D tables are
// kernel arguments: scalar loads, SGPR operands of the FMAs.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "pa_nd_hex_core.hpp"

namespace pa {

template <int P1, int Q1>
struct NDArgs {
  int ne;
  // E / E^T run in "sorted order": entry m of an element is its m-th smallest global dof, so the
  // lanes of one load / store instruction touch neighbouring dofs (few cache lines) instead of one
  // line per lane; perm[m] is the tensor-order slot of entry m (staging through LDS).
  const int32_t *sidx;     // signed sorted index (scatter side, atomic form only)
  const int32_t *sidx_in;  // the same with kEssBit set on dofs to be read as zero (gather side)
  const uint16_t *perm;    // [ne][P] tensor-order slot of sorted entry m
  const double *geom;
  const double *qdata;  // packed symmetric D, [ne][NG][Q] (QD == true)
  const double *x;
  double *y;   // L-vector target of the atomic scatter (EVEC == false)
  double *ye;  // E-vector target [ne][P], sorted order, unsigned (EVEC == true)
  // second right-hand side (NRHS == 2: y1 = A x1 in the same pass over the index and q-data streams)
  const double *x1;
  double *y1, *ye1;
  int direct;      // EVEC: entries flagged kExclBit16 in perm go straight to y (they are the only copy)
  int accumulate;  // for those: y += v instead of y = v
  int xcd_chunk;   // > 0: workgroups are dealt to the 8 XCDs round-robin; give each XCD one contiguous element range
  int ess_policy;  // -1, or ParOperator's row fix-up fused in: y[ess] = x[ess] (1) / 0 (0) (rap.cpp:223-233)
  int cross = 0;   // 1 / 2: the mixed curl forms of hcurlhdiv_33_qf.h (matrix-free D, coefficient in c_mass)
  CoeffDev c_mass, c_curl;
  NDTab<P1, Q1> tab;
  const int32_t *attr_e;     // metric form: element attributes
#ifdef PA_ABLATION
  int dbg;  // timing experiments only: 1 no E-vector store, 4 no q-data/geometry loads, 8 no gather
#endif
};


// ISO: every material coefficient is a multiple of the identity (checked at creation), so D needs
// one scalar per context instead of a 3x3 matrix.
// QD: D is read as packed symmetric matrices (pre-assembled q-data) instead of being rebuilt from
// the geometry factors: 6 (12) doubles per point instead of 11, and 9 (18) FMAs instead of ~60.
// DIRECT (only with EVEC && QD): entries flagged in `perm` are the only copy of their dof and are stored
// straight into y.
// NRHS == 2: two input vectors share one pass over the element's index arrays and D-stage data (the complex
// operator's Ar (xr, xi) / Ai (xr, xi) pairs, linalg/operator.cpp:98-134); the pipeline runs twice per element.
template <int P1, int Q1, bool USE_U, bool USE_C, bool ISO, bool EVEC, bool QD, bool DIRECT = false, int NRHS = 1>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void nd_hex_apply_kernel(const NDArgs<P1, Q1> a) {
  using L = NDLayout<P1, Q1>;
  constexpr int Q = Q1 * Q1 * Q1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / L::T, t = lane - sub * L::T;
  const int ta = t % Q1, tb = t / Q1;
  const bool lane_ok = sub < L::EPW;
  const int bid = a.xcd_chunk > 0 ? (int)(blockIdx.x & 7) * a.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int e = (bid * kWavesPerBlock + wave) * L::EPW + sub;
  const bool active = lane_ok && e < a.ne;
  double *sm = smem + (size_t)(wave * L::EPW + (lane_ok ? sub : 0)) * L::ELEM_PAD;
  const int lx = L::parity_xor(sub);  // swizzled layouts: odd elements use the other half of the banks

  // Geometry (or packed q-data) of this lane's Q1 quadrature points: issue the loads now (the
  // dominant HBM stream) and consume them after the forward contraction.
  // (Matrix-free D at Q1 >= 5 loads its geometry point by point inside the D loop instead: holding
  // 50 doubles across the contraction made those instantiations spill hundreds of registers.)
  constexpr bool METRIC = QD && ISO;  // q-data = G = J^T J (6 per point), coefficients applied here
  constexpr int NG = QD ? (METRIC ? (USE_U ? 7 : 6) : (USE_U ? 6 : 0) + (USE_C ? 6 : 0)) : 10;
  // two right-hand sides, curl-curl + mass in the metric form: the q-data is read again per point for each right-hand
  // side (the second time from L2) instead of being held, which keeps the kernel free of register spills
  constexpr bool LATEQ = METRIC && USE_U && USE_C && NRHS > 1;
  constexpr bool LATE = (!QD && Q1 >= 5) || LATEQ;
  double gd[LATE ? 1 : Q1][NG];
  int attr[LATE ? 1 : Q1];
  const double *glate = a.geom + (size_t)(active ? e : 0) * 11 * Q + ta + Q1 * tb;
  // (Q1 == 4: the two points (qz, qz + 1) of a lane's column are stored side by side, nd_qd_offset)
  const double *gq_late = a.qdata + (size_t)(active ? e : 0) * 7 * nd_qd_cstride(Q1);
  if (QD && !LATEQ) {
    const double *g = a.qdata + (size_t)(active ? e : 0) * (METRIC ? 7 : NG) * nd_qd_cstride(Q1);
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      constexpr int gs = LATE ? 0 : 1;
      attr[gs * qz] = 0;
#pragma unroll
      for (int c = 0; c < NG; c++) {
#ifdef PA_ABLATION
        if (a.dbg & 4) {
          gd[gs * qz][c] = 1.0 + 0.01 * c + 1e-3 * lane;
          continue;
        }
#endif
        gd[gs * qz][c] = g[nd_qd_offset(Q1, c, ta + Q1 * tb + Q1 * Q1 * qz)];
      }
    }
  } else if (!LATE) {
    const double *g = glate;
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      constexpr int gs = LATE ? 0 : 1;
      attr[gs * qz] = (int)g[Q1 * Q1 * qz];
#pragma unroll
      for (int c = 0; c < 10; c++) gd[gs * qz][c] = g[(1 + c) * Q + Q1 * Q1 * qz];
    }
  }

  // E: sorted-order gather, staged through LDS into tensor order
  constexpr int NC = P1 + 1, PP = 3 * P1 * NC * NC, NPL = (PP + L::T - 1) / L::T;
  int lp[NPL];
  int sg[NRHS > 1 ? NPL : 1];  // NRHS > 1: the index words are loaded once and kept
#pragma unroll
  for (int r = 0; r < NPL; r++) {
    const int m = t + L::T * r;
    lp[r] = 0;
    if (NRHS > 1) sg[r] = 0;
    if (active && m < PP) {
      lp[r] = a.perm[(size_t)e * PP + m];  // may carry kExclBit16 (used by the store below)
      if (NRHS > 1) sg[r] = a.sidx_in[(size_t)e * PP + m];
    }
  }
#pragma unroll  // straight-line copies: the table operands are re-read from the kernel arguments, not held in a loop
  for (int rhs = 0; rhs < NRHS; rhs++) {
  const double *xin = (NRHS > 1 && rhs) ? a.x1 : a.x;
  double *yout = (NRHS > 1 && rhs) ? a.y1 : a.y;
  double *yeout = (NRHS > 1 && rhs) ? a.ye1 : a.ye;
#pragma unroll
  for (int r = 0; r < NPL; r++) {
    const int m = t + L::T * r;
    if (active && m < PP) {
      const int s = (NRHS > 1) ? sg[r] : a.sidx_in[(size_t)e * PP + m];
      const int d = s >= 0 ? s : -1 - s;
      // essential dofs are flagged in the gather index: read as zero (ParOperator's tx[ess] = 0)
#ifdef PA_ABLATION
      const double xv = (a.dbg & 8) ? (double)d : ((d & kEssBit) ? 0.0 : xin[d & ~kEssBit]);
#else
      const double xv = (d & kEssBit) ? 0.0 : xin[d & ~kEssBit];
#endif
      sm[DIRECT ? (lp[r] & (kExclBit16 - 1)) : lp[r]] = s >= 0 ? xv : -xv;
    }
  }
  wave_sync();
  double uin[3][NC];
#pragma unroll
  for (int C = 0; C < 3; C++) {
    const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
    const bool act = ta < nj && tb < nk;
#pragma unroll
    for (int i = 0; i < NC; i++)
      uin[C][i] = (act && i < ni) ? sm[C * P1 * NC * NC + i + ni * (ta + nj * tb)] : 0.0;
  }
  wave_sync();

  double U[3][Q1], CU[3][Q1];
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int q = 0; q < Q1; q++) U[c][q] = 0.0, CU[c][q] = 0.0;

  nd_fwd_comp<0, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[0], U, CU);
  nd_fwd_comp<1, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[1], U, CU);
  nd_fwd_comp<2, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[2], U, CU);

  const int attr_m = METRIC ? a.attr_e[active ? e : 0] : 1;
  // D at the Q1 points of this lane's column (hcurl_33 / hdiv_33 / hdivmass_33)
#pragma unroll
  for (int qz = 0; qz < Q1; qz++) {
    constexpr int gq = LATE ? 0 : 1;  // index stride into the (pre)loaded geometry / q-data
    if (METRIC) {
      // q-data = H = (w / |detJ|) J^T J {00, 01, 02, 11, 12, 22} and, for the mass part, |detJ| / w:
      //   (w / detJ) J^T c J = c H,   w detJ adj^T c adj = c (|detJ| / w) adj(H)
      if (LATEQ) {
#pragma unroll
        for (int c = 0; c < NG; c++) gd[0][c] = gq_late[nd_qd_offset(Q1, c, ta + Q1 * tb + Q1 * Q1 * qz)];
      }
      const double *H = &gd[gq * qz][0];
      if (USE_U) {
        const double cm = gd[gq * qz][6] * a.c_mass.mat[9 * coeff_index(a.c_mass, attr_m)];
        const double m[6] = {cm * (H[3] * H[5] - H[4] * H[4]), cm * (H[2] * H[4] - H[1] * H[5]), cm * (H[1] * H[4] - H[2] * H[3]),
                             cm * (H[0] * H[5] - H[2] * H[2]), cm * (H[1] * H[2] - H[0] * H[4]), cm * (H[0] * H[3] - H[1] * H[1])};
        sym_mv(m, U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
      }
      if (USE_C) {
        const double cc = a.c_curl.mat[9 * coeff_index(a.c_curl, attr_m)];
        const double m[6] = {cc * H[0], cc * H[1], cc * H[2], cc * H[3], cc * H[4], cc * H[5]};
        sym_mv(m, CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
      }
      __builtin_amdgcn_sched_barrier(0);  // one point at a time: keeps the live range of the temporaries short
      continue;
    }
    if (QD) {
      if (USE_U) sym_mv(&gd[gq * qz][0], U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
      if (USE_C)
        sym_mv(&gd[gq * qz][USE_U ? 6 : 0], CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
      continue;
    }
    if (LATE) {
      attr[0] = (int)glate[Q1 * Q1 * qz];
#pragma unroll
      for (int c = 0; c < 10; c++) gd[0][c] = glate[(1 + c) * Q + Q1 * Q1 * qz];
    }
    const double wdetJ = gd[gq * qz][0];
    const double *adj = &gd[gq * qz][1];
    if (ISO) {
      if (USE_U) {
        const double c = a.c_mass.mat[9 * coeff_index(a.c_mass, attr[gq * qz])];
        mult_AtAx33(adj, U[0][qz], U[1][qz], U[2][qz], wdetJ * c, U[0][qz], U[1][qz], U[2][qz]);
      }
      if (USE_C) {
        double Jl[9];
        const double c = a.c_curl.mat[9 * coeff_index(a.c_curl, attr[gq * qz])];
        adjJt33(adj, Jl);
        mult_AtAx33(Jl, CU[0][qz], CU[1][qz], CU[2][qz], wdetJ * c, CU[0][qz], CU[1][qz], CU[2][qz]);
      }
    } else if (USE_U && USE_C && a.cross) {
      // hcurlhdiv_33_qf.h: cross 1 = f_apply_hcurlhdiv_33 (values in, curl test functions out), 2 = f_apply_hdivhcurl_33
      double Cm[9], Jl[9], o0, o1, o2;
      coeff_unpack3(a.c_mass, attr[gq * qz], Cm);
      adjJt33(adj, Jl);
      if (a.cross == 1) {
        mult_AtBCx33(Jl, Cm, adj, U[0][qz], U[1][qz], U[2][qz], wdetJ, o0, o1, o2);
        CU[0][qz] = o0, CU[1][qz] = o1, CU[2][qz] = o2;
        U[0][qz] = U[1][qz] = U[2][qz] = 0.0;
      } else {
        mult_AtBCx33(adj, Cm, Jl, CU[0][qz], CU[1][qz], CU[2][qz], wdetJ, o0, o1, o2);
        U[0][qz] = o0, U[1][qz] = o1, U[2][qz] = o2;
        CU[0][qz] = CU[1][qz] = CU[2][qz] = 0.0;
      }
    } else {
      double Cm[9];
      if (USE_U) {
        coeff_unpack3(a.c_mass, attr[gq * qz], Cm);
        mult_AtBCx33(adj, Cm, adj, U[0][qz], U[1][qz], U[2][qz], wdetJ, U[0][qz], U[1][qz], U[2][qz]);
      }
      if (USE_C) {
        double Jl[9];
        coeff_unpack3(a.c_curl, attr[gq * qz], Cm);
        adjJt33(adj, Jl);
        mult_AtBCx33(Jl, Cm, Jl, CU[0][qz], CU[1][qz], CU[2][qz], wdetJ, CU[0][qz], CU[1][qz],
                     CU[2][qz]);
      }
    }
  }

  nd_bwd_comp<0, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[0], U, CU);
  nd_bwd_comp<1, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[1], U, CU);
  nd_bwd_comp<2, P1, Q1, USE_U, USE_C>(a, e, active, lane_ok, ta, tb, lx, sm, uin[2], U, CU);

  // E^T, first half: element-local results back into tensor order in LDS, then out in sorted order
  // (coalesced E-vector store; signs and the sum over elements happen in et_gather_kernel)
#pragma unroll
  for (int C = 0; C < 3; C++) {
    const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
    const bool act = lane_ok && ta < nj && tb < nk;
#pragma unroll
    for (int i = 0; i < NC; i++)
      if (act && i < ni) sm[C * P1 * NC * NC + i + ni * (ta + nj * tb)] = uin[C][i];
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < NPL; r++) {
    const int m = t + L::T * r;
    if (active && m < PP) {
      const double v = sm[DIRECT ? (lp[r] & (kExclBit16 - 1)) : lp[r]];
#ifdef PA_ABLATION
      if (a.dbg & 1) {
        asm volatile("" ::"v"(v));
        continue;
      }
#endif
      if (EVEC) {
        if (DIRECT && (lp[r] & kExclBit16)) {  // only copy of this dof: no E-vector round trip, no gather
          const int s = (NRHS > 1) ? sg[r] : a.sidx_in[(size_t)e * PP + m];  // carries kEssBit on essential dofs when masked
          const int df = s >= 0 ? s : -1 - s, d = df & ~kEssBit;
          double *dst = &yout[d];
          const double sv = s >= 0 ? v : -v;
          if ((df & kEssBit) && a.ess_policy >= 0)
            *dst = a.ess_policy ? xin[d] : 0.0;
          else
            *dst = a.accumulate ? *dst + sv : sv;
        } else {
          yeout[(size_t)e * PP + m] = v;
        }
      } else {
        const int s = a.sidx[(size_t)e * PP + m];
        unsafeAtomicAdd(&yout[s >= 0 ? s : -1 - s], s >= 0 ? v : -v);
      }
    }
  }
  if (NRHS > 1) wave_sync();  // the LDS strip is reused by the next right-hand side
  }  // rhs
}

template <int P1, int Q1>
static void fill_tab(const SubOp &so, NDTab<P1, Q1> &t) {
  // the kept entries are a prefix of the full row-major tables
  for (int i = 0; i < HalfTab<P1, Q1>::LEN; i++) t.Bo[i] = so.Bo[i];
  for (int i = 0; i < HalfTab<P1 + 1, Q1>::LEN; i++) t.Bc[i] = so.Bc[i], t.Gc[i] = so.Gc[i];
}

// The exclusive-dof store is used by the q-data kernels with Q1 <= 4 (the Q1 = 5 instantiations are
// short of SGPRs as it is); the gather kernel must make the same choice.
static bool use_direct(const SubOp &so) { return so.d_perm_x && so.d_shared && so.qd && so.q1d <= 4; }

template <int P1, int Q1, bool U, bool C>
static void launch_iso(const NDArgs<P1, Q1> &a, bool iso, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  const bool evec = a.ye != nullptr;
  if (a.x1) {  // two right-hand sides (q-data forms with the E-vector, Q1 <= 4; the caller checks)
    if (a.attr_e) {
      if (a.direct)
        hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, true, true, (Q1 <= 4), (Q1 <= 4 ? 2 : 1)>), grid, block, lds, s, a);
      else
        hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, true, true, false, (Q1 <= 4 ? 2 : 1)>), grid, block, lds, s, a);
    } else {
      if (a.direct)
        hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, true, true, (Q1 <= 4), (Q1 <= 4 ? 2 : 1)>), grid, block, lds, s, a);
      else
        hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, true, true, false, (Q1 <= 4 ? 2 : 1)>), grid, block, lds, s, a);
    }
    return;
  }
  if (a.qdata && a.attr_e) {  // metric form of the q-data
    if (evec && a.direct && Q1 <= 4)
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, true, true, (Q1 <= 4)>), grid, block, lds, s, a);
    else if (evec)
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, true, true>), grid, block, lds, s, a);
    else
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, false, true>), grid, block, lds, s, a);
  } else if (a.qdata) {
    if (evec && a.direct && Q1 <= 4)
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, true, true, (Q1 <= 4)>), grid, block, lds, s, a);
    else if (evec)
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, true, true>), grid, block, lds, s, a);
    else
      hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, false, true>), grid, block, lds, s, a);
  } else if (iso && evec)
    hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, true, false>), grid, block, lds, s, a);
  else if (iso)
    hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, true, false, false>), grid, block, lds, s, a);
  else if (evec)
    hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, true, false>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((nd_hex_apply_kernel<P1, Q1, U, C, false, false, false>), grid, block, lds, s, a);
}

template <int P1, int Q1>
static void launch_pq(const SubOp &so, const double *x, double *y, double *ye, bool masked, hipStream_t s,
                      bool accumulate, int ess_policy, const double *x1, double *y1, double *ye1) {
  using L = NDLayout<P1, Q1>;
  NDArgs<P1, Q1> a;
  a.ne = so.ne;
  a.sidx = so.d_sidx;
  a.sidx_in = (masked && so.d_sidx_bc) ? so.d_sidx_bc : so.d_sidx;
  a.direct = (ye != nullptr && use_direct(so)) ? 1 : 0;
  a.perm = a.direct ? so.d_perm_x : so.d_perm;
  a.accumulate = accumulate ? 1 : 0;
  a.ess_policy = (a.direct && masked) ? ess_policy : -1;
  a.geom = so.geom->d_geom;
  a.qdata = so.qd ? so.qd->d : nullptr;
  a.attr_e = nullptr;
  if (so.qd && so.qd->metric) a.attr_e = so.geom->d_attr_e;
  a.x = x;
  a.y = y;
  a.ye = ye;
  a.x1 = x1, a.y1 = y1, a.ye1 = ye1;
  fill_tab(so, a.tab);
#ifdef PA_ABLATION
  a.dbg = getenv("PA_DBG") ? atoi(getenv("PA_DBG")) : 0;
#endif
  const int epb = kWavesPerBlock * L::EPW;
  const int nblk = (so.ne + epb - 1) / epb;
  static const bool xcd_map = !(getenv("PALACE_AMD_XCD") && atoi(getenv("PALACE_AMD_XCD")) == 0);
  a.xcd_chunk = (xcd_map && nblk >= 64) ? (nblk + 7) / 8 : 0;
  const dim3 grid(a.xcd_chunk > 0 ? 8 * a.xcd_chunk : nblk), block(64 * kWavesPerBlock);
  const size_t lds = sizeof(double) * (size_t)epb * L::ELEM_PAD;
  switch (so.qf) {
    case PA_QF_HDIV_33:
      a.c_curl = so.c0.dev();
      launch_iso<P1, Q1, false, true>(a, so.iso, grid, block, lds, s);
      break;
    case PA_QF_HCURL_33:
      a.c_mass = so.c0.dev();
      launch_iso<P1, Q1, true, false>(a, so.iso, grid, block, lds, s);
      break;
    case PA_QF_HDIVMASS_33:
      a.c_mass = so.c0.dev();
      a.c_curl = so.c1.dev();
      launch_iso<P1, Q1, true, true>(a, so.iso, grid, block, lds, s);
      break;
    case PA_QF_HCURLHDIV_33:
    case PA_QF_HDIVHCURL_33:
      // both fields are evaluated, D couples values and curls (matrix-free, general 3 x 3 coefficient); the transposed
      // operator of one form is the other one with the transposed coefficient (TransposeScope swaps the latter)
      a.c_mass = a.c_curl = so.c0.dev();
      a.cross = ((so.qf == PA_QF_HCURLHDIV_33) != TransposeScope::active()) ? 1 : 2;
      launch_iso<P1, Q1, true, true>(a, false, grid, block, lds, s);
      break;
    default:
      throw Error("QFunction not available for H(curl) hexahedra");
  }
  PA_HIP(hipGetLastError());
}

#define PA_ND_DISPATCH(FN, ...)                                                           \
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
      throw Error("no H(curl) hex kernel for order " + std::to_string(so.p) + " with " +   \
                  std::to_string(so.q1d) + " points per direction");                       \
  }

// ye != nullptr: write the element-local results (E-vector) instead of scattering atomically into y
// masked: gather through the essential-dof-flagged index array (pa_op_set_essential)
bool nd_hex_supports_two_rhs(const SubOp &so) { return so.fe_type == PA_FE_HCURL && so.qd && so.d_ye && so.q1d <= 4; }

void launch_nd_hex_apply(const SubOp &so, const double *x, double *y, double *ye, bool masked, hipStream_t s,
                         bool accumulate, int ess_policy, const double *x1, double *y1, double *ye1) {
  PA_ND_DISPATCH(launch_pq, so, x, y, ye, masked, s, accumulate, ess_policy, x1, y1, ye1)
}

// ---- E^T as a gather: y_d (+)= sum over the element-local copies of dof d -----------------------
// tptr/tent: transpose of the signed tensor-order index array (CSR by L-dof); entry t >= 0 reads
// ye[t], t < 0 reads -ye[-1-t].  One thread per dof, fixed summation order => reproducible.
__global__ void et_gather_kernel(const int n, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tent,
                                 const double *__restrict__ ye, double *__restrict__ y, const int accumulate,
                                 const int32_t *__restrict__ list, const double *__restrict__ x, const int ess_policy) {
  const int k0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k0 >= n) return;
  const int dl = list ? list[k0] : k0;  // list: only the dofs with more than one copy (kEssBit: essential)
  const int d = dl & ~kEssBit;
  if ((dl & kEssBit) && ess_policy >= 0) {  // ParOperator's essential rows (rap.cpp:223-233), fused
    y[d] = ess_policy ? x[d] : 0.0;
    return;
  }
  const int b = tptr[d], e = tptr[d + 1];
  double s = 0.0;
  int k = b;
  // dofs have 1 (interior), 2 (face) or ~4 (edge) copies: fetch four at a time so the loads overlap
  for (; k + 4 <= e; k += 4) {
    const int t0 = tent[k], t1 = tent[k + 1], t2 = tent[k + 2], t3 = tent[k + 3];
    const double v0 = ye[t0 >= 0 ? t0 : -1 - t0], v1 = ye[t1 >= 0 ? t1 : -1 - t1];
    const double v2 = ye[t2 >= 0 ? t2 : -1 - t2], v3 = ye[t3 >= 0 ? t3 : -1 - t3];
    s += t0 >= 0 ? v0 : -v0;
    s += t1 >= 0 ? v1 : -v1;
    s += t2 >= 0 ? v2 : -v2;
    s += t3 >= 0 ? v3 : -v3;
  }
  if (k + 2 <= e) {
    const int t0 = tent[k], t1 = tent[k + 1];
    const double v0 = ye[t0 >= 0 ? t0 : -1 - t0], v1 = ye[t1 >= 0 ? t1 : -1 - t1];
    s += t0 >= 0 ? v0 : -v0;
    s += t1 >= 0 ? v1 : -v1;
    k += 2;
  }
  if (k < e) {
    const int t0 = tent[k];
    const double v0 = ye[t0 >= 0 ? t0 : -1 - t0];
    s += t0 >= 0 ? v0 : -v0;
  }
  y[d] = accumulate ? y[d] + s : s;
}

// the same for two E-vector / y pairs with one read of the transpose map
__global__ void et_gather2_kernel(const int n, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tent,
                                  const double *__restrict__ ye0, const double *__restrict__ ye1, double *__restrict__ y0,
                                  double *__restrict__ y1, const int accumulate, const int32_t *__restrict__ list,
                                  const double *__restrict__ x0, const double *__restrict__ x1, const int ess_policy) {
  const int k0 = blockIdx.x * blockDim.x + threadIdx.x;
  if (k0 >= n) return;
  const int dl = list ? list[k0] : k0;
  const int d = dl & ~kEssBit;
  if ((dl & kEssBit) && ess_policy >= 0) {
    y0[d] = ess_policy ? x0[d] : 0.0;
    y1[d] = ess_policy ? x1[d] : 0.0;
    return;
  }
  double s0 = 0.0, s1 = 0.0;
  for (int k = tptr[d]; k < tptr[d + 1]; k++) {
    const int t = tent[k];
    const int u = t >= 0 ? t : -1 - t;
    const double v0 = ye0[u], v1 = ye1[u];
    s0 += t >= 0 ? v0 : -v0;
    s1 += t >= 0 ? v1 : -v1;
  }
  y0[d] = accumulate ? y0[d] + s0 : s0;
  y1[d] = accumulate ? y1[d] + s1 : s1;
}

void launch_et_gather2(const SubOp &so, double *y0, double *y1, bool accumulate, hipStream_t s, const double *x0,
                       const double *x1, int ess_policy) {
  const bool dir = use_direct(so);
  const int n = dir ? so.n_shared : so.lsize;
  const int32_t *list = dir ? ((ess_policy >= 0 && so.d_shared_bc) ? so.d_shared_bc : so.d_shared) : nullptr;
  if (n == 0) return;
  hipLaunchKernelGGL(et_gather2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, so.d_tptr, so.d_tent, so.d_ye, so.d_ye2, y0,
                     y1, accumulate ? 1 : 0, list, x0, x1, (dir && so.d_shared_bc) ? ess_policy : -1);
  PA_HIP(hipGetLastError());
}

void launch_et_gather_raw(int n, const int32_t *tptr, const int32_t *tent, const double *ye, double *y,
                          bool accumulate, hipStream_t s, const int32_t *list, const double *x, int ess_policy) {
  const int bs = 256;
  if (n == 0) return;
  hipLaunchKernelGGL(et_gather_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, s, n, tptr, tent, ye, y,
                     accumulate ? 1 : 0, list, x, ess_policy);
  PA_HIP(hipGetLastError());
}

bool nd_hex_fuses_essential(const SubOp &so) {
  // (five points per direction: only the streaming kernel's run gather owns the essential rows, pa_nd_hex_stream5.hip)
  return (use_direct(so) && so.d_shared_bc != nullptr) || (so.q1d == 5 && so.d_idxc && so.d_perm_s_bc);
}

void launch_et_gather(const SubOp &so, double *y, bool accumulate, hipStream_t s, const double *x, int ess_policy) {
  if (use_direct(so) && ess_policy >= 0 && so.d_shared_bc)
    launch_et_gather_raw(so.n_shared, so.d_tptr, so.d_tent, so.d_ye, y, accumulate, s, so.d_shared_bc, x, ess_policy);
  else if (use_direct(so))  // the element kernel stored the exclusive dofs itself
    launch_et_gather_raw(so.n_shared, so.d_tptr, so.d_tent, so.d_ye, y, accumulate, s, so.d_shared);
  else
    launch_et_gather_raw(so.lsize, so.d_tptr, so.d_tent, so.d_ye, y, accumulate, s);
}

// ---- packed symmetric q-data (set-up) -----------------------------------------------------------
// The reference can pre-assemble D too (assemble_q_data, fem/libceed/integrator.cpp:158-314, stores
// the full 3x3); here the symmetric matrices are stored packed.  One thread per point.
__global__ void nd_hex_qdata_kernel(const int ne, const int Q, const int q1d, const double *__restrict__ geom, const CoeffDev c_mass,
                                    const CoeffDev c_curl, const int use_u, const int use_c, double *__restrict__ qd) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  const double *g = geom + (size_t)e * 11 * Q;
  const int ncomp = 6 * (use_u + use_c);
  double *out = qd + (size_t)e * ncomp * nd_qd_cstride(q1d);
  double adj[9], Cm[9], Jl[9], M[9];
  const int attr = (int)g[q];
  const double w = g[Q + q];
  for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + q];
  int o = 0;
  for (int part = 0; part < 2; part++) {
    if ((part == 0 && !use_u) || (part == 1 && !use_c)) continue;
    if (part == 0) {
      coeff_unpack3(c_mass, attr, Cm);
      for (int c = 0; c < 9; c++) Jl[c] = adj[c];
    } else {
      coeff_unpack3(c_curl, attr, Cm);
      adjJt33(adj, Jl);
    }
    for (int col = 0; col < 3; col++)
      mult_AtBCx33(Jl, Cm, Jl, col == 0, col == 1, col == 2, w, M[0 + 3 * col], M[1 + 3 * col], M[2 + 3 * col]);
    // symmetric by construction when C is; average the off-diagonal pairs against rounding drift
    out[nd_qd_offset(q1d, o + 0, q)] = M[0];
    out[nd_qd_offset(q1d, o + 1, q)] = 0.5 * (M[3] + M[1]);
    out[nd_qd_offset(q1d, o + 2, q)] = 0.5 * (M[6] + M[2]);
    out[nd_qd_offset(q1d, o + 3, q)] = M[4];
    out[nd_qd_offset(q1d, o + 4, q)] = 0.5 * (M[7] + M[5]);
    out[nd_qd_offset(q1d, o + 5, q)] = M[8];
    o += 6;
  }
}

void launch_nd_hex_qdata(SubOp &so, hipStream_t s) {
  const bool use_u = so.qf == PA_QF_HCURL_33 || so.qf == PA_QF_HDIVMASS_33;
  const bool use_c = so.qf == PA_QF_HDIV_33 || so.qf == PA_QF_HDIVMASS_33;
  auto *qd = new QData;
  qd->ncomp = 6 * ((int)use_u + (int)use_c);
  // (padded to a multiple of four elements for the streaming kernel; the pad is never used in a result)
  const size_t nq = (size_t)((so.ne + 3) & ~3) * qd->ncomp * nd_qd_cstride(so.q1d);
  qd->d = dev_alloc<double>(nq);
  PA_HIP(hipMemsetAsync(qd->d, 0, nq * sizeof(double), s));
  CoeffDev cm{}, cc{};
  if (so.qf == PA_QF_HDIV_33) cc = so.c0.dev();
  if (so.qf == PA_QF_HCURL_33) cm = so.c0.dev();
  if (so.qf == PA_QF_HDIVMASS_33) cm = so.c0.dev(), cc = so.c1.dev();
  const long long n = (long long)so.ne * so.Q;
  hipLaunchKernelGGL(nd_hex_qdata_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, so.ne, so.Q, so.q1d,
                     so.geom->d_geom, cm, cc, (int)use_u, (int)use_c, qd->d);
  PA_HIP(hipGetLastError());
  so.qd = qd;
}

// Metric q-data from the geometry factors: adjJt33(adj) = J / detJ (the trick hdiv_33_qf.h uses), detJ =
// (w detJ) / w; stored per point: H = (w / |detJ|) J^T J (six entries) and |detJ| / w.  One thread per point.
__global__ void nd_hex_metric_kernel(const int ne, const int q1d, const double *__restrict__ geom,
                                     const double *__restrict__ w1, double *__restrict__ qd) {
  const int Q = q1d * q1d * q1d;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  const double *g = geom + (size_t)e * 11 * Q;
  double adj[9], Jl[9];
  for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + q];
  adjJt33(adj, Jl);
  const double w = w1[q % q1d] * w1[(q / q1d) % q1d] * w1[q / (q1d * q1d)];
  const double det = g[Q + q] / w;
  const double k = w * fabs(det);  // (w / |det|) (det Jl)^T (det Jl) = w |det| Jl^T Jl
  double *out = qd + (size_t)e * 7 * nd_qd_cstride(q1d);
  int o = 0;
  for (int i = 0; i < 3; i++)
    for (int j = i; j < 3; j++)
      out[nd_qd_offset(q1d, o++, q)] = k * (Jl[3 * i] * Jl[3 * j] + Jl[3 * i + 1] * Jl[3 * j + 1] + Jl[3 * i + 2] * Jl[3 * j + 2]);
  out[nd_qd_offset(q1d, 6, q)] = fabs(det) / w;
}

void launch_nd_hex_metric(SubOp &so, hipStream_t s) {
  Geom &g = *so.geom;
  auto *qd = new QData;
  qd->ncomp = 7, qd->metric = true;
  const size_t nq = (size_t)((so.ne + 3) & ~3) * 7 * nd_qd_cstride(so.q1d);
  qd->d = dev_alloc<double>(nq);
  PA_HIP(hipMemsetAsync(qd->d, 0, nq * sizeof(double), s));
  double *d_w = dev_upload(g.w1.data(), g.w1.size(), s);
  const long long n = (long long)so.ne * so.Q;
  hipLaunchKernelGGL(nd_hex_metric_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, so.ne, so.q1d, g.d_geom, d_w,
                     qd->d);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  hipFree(d_w);
  g.metric = qd;  // the geometry data holds the first reference
}

// ---- diagonal -------------------------------------------------------------------------------
// diag_l = sum_q [ phi_l^T Mm phi_l + curl(phi_l)^T Mc curl(phi_l) ] with the pointwise matrices
// Mm = w detJ adj^T C adj and Mc = w detJ Jl^T C Jl of the D stage.  Set-up only (reference
// operator.cpp:116-143 / CeedOperatorLinearAssembleAddDiagonal): one workgroup per element,
// matrices staged in LDS, one thread per local dof; full 1-D tables read from device memory.
struct NDDiagArgs {
  int ne, p, q1;
  const int32_t *lidx;
  // E-vector form (default): entry m of the element in sorted order, summed per dof by et_gather_kernel in a fixed order
  const int32_t *sidx;
  const uint16_t *perm;
  double *ye;
  const double *geom;
  double *y;
  CoeffDev c_mass, c_curl;
  const double *Bo, *Bc, *Gc;  // device, full [q1][n]
  bool use_u, use_c;
  int cross;  // 1 / 2: the mixed forms (hcurlhdiv_33_qf.h), coefficient in c_mass
};

__global__ void nd_hex_diag_kernel(const NDDiagArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int P1 = a.p, Q1 = a.q1, NC = P1 + 1, Q = Q1 * Q1 * Q1, P = 3 * P1 * NC * NC;
  double *Mm = dsm, *Mc = dsm + 9 * Q;
  const int e = blockIdx.x;
  const double *g = a.geom + (size_t)e * 11 * Q;
  for (int q = threadIdx.x; q < Q; q += blockDim.x) {
    double adj[9], Cm[9], Jl[9];
    const int attr = (int)g[q];
    const double w = g[Q + q];
    for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + q];
    for (int col = 0; col < 3; col++) {
      const double e0 = col == 0, e1 = col == 1, e2 = col == 2;
      double y0 = 0, y1 = 0, y2 = 0;
      if (a.use_u) {
        coeff_unpack3(a.c_mass, attr, Cm);
        mult_AtBCx33(adj, Cm, adj, e0, e1, e2, w, y0, y1, y2);
      }
      Mm[9 * q + 0 + 3 * col] = y0, Mm[9 * q + 1 + 3 * col] = y1, Mm[9 * q + 2 + 3 * col] = y2;
      y0 = y1 = y2 = 0;
      if (a.use_c) {
        coeff_unpack3(a.c_curl, attr, Cm);
        adjJt33(adj, Jl);
        mult_AtBCx33(Jl, Cm, Jl, e0, e1, e2, w, y0, y1, y2);
      }
      Mc[9 * q + 0 + 3 * col] = y0, Mc[9 * q + 1 + 3 * col] = y1, Mc[9 * q + 2 + 3 * col] = y2;
      if (a.cross) {  // column `col` of  w detJ Jl^T C adj  (cross 1) or  w detJ adj^T C Jl  (cross 2), kept in Mm
        coeff_unpack3(a.c_mass, attr, Cm);
        adjJt33(adj, Jl);
        if (a.cross == 1)
          mult_AtBCx33(Jl, Cm, adj, e0, e1, e2, w, y0, y1, y2);
        else
          mult_AtBCx33(adj, Cm, Jl, e0, e1, e2, w, y0, y1, y2);
        Mm[9 * q + 0 + 3 * col] = y0, Mm[9 * q + 1 + 3 * col] = y1, Mm[9 * q + 2 + 3 * col] = y2;
        Mc[9 * q + 0 + 3 * col] = Mc[9 * q + 1 + 3 * col] = Mc[9 * q + 2 + 3 * col] = 0.0;
      }
    }
  }
  __syncthreads();
  for (int m = threadIdx.x; m < P; m += blockDim.x) {
    const int l = a.ye ? a.perm[(size_t)e * P + m] : m;
    const int C = l / (P1 * NC * NC);
    const int r = l - C * P1 * NC * NC;
    const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
    const int i = r % ni, j = (r / ni) % nj, k = r / (ni * nj);
    const double *TX = (C == 0) ? a.Bo : a.Bc;
    const double *TY = (C == 1) ? a.Bo : a.Bc;
    const double *TZ = (C == 2) ? a.Bo : a.Bc;
    double acc = 0.0;
    for (int qz = 0; qz < Q1; qz++)
      for (int qy = 0; qy < Q1; qy++)
        for (int qx = 0; qx < Q1; qx++) {
          const int q = qx + Q1 * (qy + Q1 * qz);
          const double bx = TX[qx * ni + i], by = TY[qy * nj + j], bz = TZ[qz * nk + k];
          const double gx = (C == 0) ? 0.0 : a.Gc[qx * NC + i];
          const double gy = (C == 1) ? 0.0 : a.Gc[qy * NC + j];
          const double gz = (C == 2) ? 0.0 : a.Gc[qz * NC + k];
          const double f = bx * by * bz;
          const double dx = gx * by * bz, dy = bx * gy * bz, dz = bx * by * gz;
          double cv[3];
          if (C == 0) cv[0] = 0.0, cv[1] = dz, cv[2] = -dy;
          if (C == 1) cv[0] = -dz, cv[1] = 0.0, cv[2] = dx;
          if (C == 2) cv[0] = dy, cv[1] = -dx, cv[2] = 0.0;
          if (a.cross == 1) {  // test: curl, trial: value f e_C
            for (int r2 = 0; r2 < 3; r2++) acc += cv[r2] * Mm[9 * q + r2 + 3 * C] * f;
          } else if (a.cross == 2) {  // test: value f e_C, trial: curl
            for (int c2 = 0; c2 < 3; c2++) acc += f * Mm[9 * q + C + 3 * c2] * cv[c2];
          } else {
            acc += Mm[9 * q + C + 3 * C] * f * f;
            for (int r2 = 0; r2 < 3; r2++)
              for (int c2 = 0; c2 < 3; c2++) acc += cv[r2] * Mc[9 * q + r2 + 3 * c2] * cv[c2];
          }
        }
    if (a.ye) {  // the gather applies the orientation sign of the entry: the diagonal does not have one
      a.ye[(size_t)e * P + m] = a.sidx[(size_t)e * P + m] >= 0 ? acc : -acc;
    } else {
      const int s = a.lidx[(size_t)e * P + l];
      unsafeAtomicAdd(&a.y[s >= 0 ? s : -1 - s], acc);
    }
  }
}

void launch_nd_hex_diag(const SubOp &so, double *diag, hipStream_t s) {
  NDDiagArgs a;
  a.ne = so.ne, a.p = so.p, a.q1 = so.q1d;
  a.lidx = so.d_lidx;
  a.sidx = so.d_sidx, a.perm = so.d_perm, a.ye = (so.d_tptr && so.d_sidx && so.d_perm) ? so.d_ye : nullptr;
  a.geom = so.geom->d_geom;
  a.y = diag;
  const int nc = so.p + 1;
  a.Bo = so.d_tab, a.Bc = so.d_tab + so.q1d * so.p, a.Gc = a.Bc + so.q1d * nc;
  a.use_u = a.use_c = false;
  a.cross = 0;
  switch (so.qf) {
    case PA_QF_HCURLHDIV_33: a.c_mass = so.c0.dev(), a.cross = 1; break;
    case PA_QF_HDIVHCURL_33: a.c_mass = so.c0.dev(), a.cross = 2; break;
    case PA_QF_HDIV_33: a.c_curl = so.c0.dev(), a.use_c = true; break;
    case PA_QF_HCURL_33: a.c_mass = so.c0.dev(), a.use_u = true; break;
    case PA_QF_HDIVMASS_33: a.c_mass = so.c0.dev(), a.c_curl = so.c1.dev(), a.use_u = a.use_c = true; break;
    default: throw Error("QFunction not available for H(curl) hexahedra");
  }
  const size_t lds = sizeof(double) * 18 * (size_t)so.Q;
  hipLaunchKernelGGL(nd_hex_diag_kernel, dim3(so.ne), dim3(128), lds, s, a);
  PA_HIP(hipGetLastError());
  // diag += sum of the element diagonals, dof by dof in a fixed order (reproducible: the smoothers' eigenvalue
  // estimates and with them the iterates do not depend on the run)
  if (a.ye) launch_et_gather_raw(so.lsize, so.d_tptr, so.d_tent, so.d_ye, diag, true, s);
}

}  // namespace pa
