// p-prolongation / restriction between Nedelec (or H1) spaces of two orders on the same hex mesh.
//
// Replaces the libCEED "interpolator" operator Palace builds in DiscreteLinearOperator::
// PartialAssemble (reference fem/bilinearform.cpp:203-282; identity QFunction + projection-matrix
// basis, fem/libceed/integrator.cpp:515-548, fem/libceed/basis.cpp:116-165) and the multiplicity
// scaling around it (fem/libceed/operator.cpp:182-240).  The element matrix is a Kronecker product
// of 1-D nodal interpolation matrices, applied by sum factorisation; mapping as in pa_nd_hex.hip:
// (p_f+1)^2 lanes per element, hand-offs through LDS inside the wave.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "linalg.hpp"
#include "comm.hpp"

namespace palace {

namespace {

constexpr int kMaxN = 6;  // closed nodes per direction

struct InterpArgs {
  int kind;  // 0: p-prolongation within one element family; 1: discrete gradient H1(p) -> ND(p)
  int ne, fe_type, pc, pf;
  const int32_t *lidx_c, *lidx_f;
  const double *Ic, *Io;      // device: [(pf+1)*(pc+1)], [pf*pc]
  const double *x;
  double *y;      // forward: fine L-vector (owner copies store, no atomics)
  double *ye_c;   // transpose: coarse E-vector [ne][Pc], summed per dof by k_gather
  double Ic_s[25], Io_s[20];  // Ic / Io by value for the specialised kernels (pf <= 4): scalar operands
};

// Every fine dof is shared by several elements that all compute the same interpolated value (the
// row of the element matrix for a shared dof only involves coarse dofs of the shared entity), so
// instead of adding all copies and scaling by 1/multiplicity (bilinearform.cpp:256-279) exactly one
// copy - the owner, flagged in the fine index array - is stored: no atomics, no memset, fixed
// result.  The transpose reads the fine vector through the same owner mask, which is the exact
// transpose of that operator.
constexpr int kOwnBit = 1 << 29;

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One component block: coarse dims (nc[0..2]) -> fine dims (nf[0..2]); M[d] the 1-D matrix
// [nf[d]][nc[d]] of direction d.  TRANSPOSE applies the transposed element matrix.
template <bool TRANSPOSE>
__device__ void interp_block(const InterpArgs &a, const int e, const bool active, const bool lane_ok,
                             const int ta, const int tb, double *sm, const bool accumulate, const int off_c,
                             const int off_f,
                             const int Pc, const int Pf, const int nc0, const int nc1, const int nc2,
                             const int nf0, const int nf1, const int nf2, const double *M0,
                             const double *M1, const double *M2) {
  const int n1c = a.pf + 1;
  double *sA = sm, *sB = sm + n1c * n1c * n1c;
  if (!TRANSPOSE) {
    // pass X: lane (j_c, k_c) -> fine i
    {
      const bool act = ta < nc1 && tb < nc2;
      double u[kMaxN];
      for (int i = 0; i < nc0; i++) {
        double v = 0.0;
        if (active && act) {
          const int s = a.lidx_c[(size_t)e * Pc + off_c + i + nc0 * (ta + nc1 * tb)];
          const double xv = a.x[s >= 0 ? s : -1 - s];
          v = s >= 0 ? xv : -xv;
        }
        u[i] = v;
      }
      for (int fi = 0; fi < nf0; fi++) {
        double v = 0.0;
        for (int i = 0; i < nc0; i++) v += M0[fi * nc0 + i] * u[i];
        if (lane_ok && act) sA[(fi * nc1 + ta) * nc2 + tb] = v;
      }
    }
    wsync();
    // pass Y: lane (i_f, k_c) -> fine j
    {
      const bool act = ta < nf0 && tb < nc2;
      double u[kMaxN];
      for (int j = 0; j < nc1; j++) u[j] = sA[((act ? ta : 0) * nc1 + j) * nc2 + (act ? tb : 0)];
      for (int fj = 0; fj < nf1; fj++) {
        double v = 0.0;
        for (int j = 0; j < nc1; j++) v += M1[fj * nc1 + j] * u[j];
        if (lane_ok && act) sB[(ta * nf1 + fj) * nc2 + tb] = v;
      }
    }
    wsync();
    // pass Z: lane (i_f, j_f) -> fine k, scale by 1/multiplicity, scatter-add
    {
      const bool act = ta < nf0 && tb < nf1;
      double u[kMaxN];
      for (int k = 0; k < nc2; k++) u[k] = sB[((act ? ta : 0) * nf1 + (act ? tb : 0)) * nc2 + k];
      for (int fk = 0; fk < nf2; fk++) {
        double v = 0.0;
        for (int k = 0; k < nc2; k++) v += M2[fk * nc2 + k] * u[k];
        if (active && act) {
          const int s = a.lidx_f[(size_t)e * Pf + off_f + ta + nf0 * (tb + nf1 * fk)];
          const int g = s >= 0 ? s : -1 - s;
          if (g & kOwnBit) a.y[g & ~kOwnBit] = s >= 0 ? v : -v;
        }
      }
    }
    wsync();
  } else {
    // pass Z^T: lane (i_f, j_f): gather fine (scaled), contract fine k -> coarse k
    {
      const bool act = ta < nf0 && tb < nf1;
      double u[kMaxN];
      for (int fk = 0; fk < nf2; fk++) {
        double v = 0.0;
        if (active && act) {
          const int s = a.lidx_f[(size_t)e * Pf + off_f + ta + nf0 * (tb + nf1 * fk)];
          const int g = s >= 0 ? s : -1 - s;
          const double xv = (g & kOwnBit) ? a.x[g & ~kOwnBit] : 0.0;
          v = s >= 0 ? xv : -xv;
        }
        u[fk] = v;
      }
      for (int k = 0; k < nc2; k++) {
        double v = 0.0;
        for (int fk = 0; fk < nf2; fk++) v += M2[fk * nc2 + k] * u[fk];
        if (lane_ok && act) sB[(ta * nf1 + tb) * nc2 + k] = v;
      }
    }
    wsync();
    // pass Y^T: lane (i_f, k_c): contract fine j -> coarse j
    {
      const bool act = ta < nf0 && tb < nc2;
      double u[kMaxN];
      for (int fj = 0; fj < nf1; fj++) u[fj] = sB[((act ? ta : 0) * nf1 + fj) * nc2 + (act ? tb : 0)];
      for (int j = 0; j < nc1; j++) {
        double v = 0.0;
        for (int fj = 0; fj < nf1; fj++) v += M1[fj * nc1 + j] * u[fj];
        if (lane_ok && act) sA[(ta * nc1 + j) * nc2 + tb] = v;
      }
    }
    wsync();
    // pass X^T: lane (j_c, k_c): contract fine i -> coarse i, scatter-add
    {
      const bool act = ta < nc1 && tb < nc2;
      double u[kMaxN];
      for (int fi = 0; fi < nf0; fi++) u[fi] = sA[(fi * nc1 + (act ? ta : 0)) * nc2 + (act ? tb : 0)];
      for (int i = 0; i < nc0; i++) {
        double v = 0.0;
        for (int fi = 0; fi < nf0; fi++) v += M0[fi * nc0 + i] * u[fi];
        if (active && act) {
          // coarse E-vector (unsigned); the three components of a gradient hit the same entries
          // from the same lane, one after the other
          double *dst = &a.ye_c[(size_t)e * Pc + off_c + i + nc0 * (ta + nc1 * tb)];
          *dst = accumulate ? *dst + v : v;
        }
      }
    }
    wsync();
  }
}

template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void interp_kernel(const InterpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int n1 = a.pf + 1, T = n1 * n1, EPW = 64 / T;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / T, t = lane - sub * T;
  const int ta = t % n1, tb = t / n1;
  const bool lane_ok = sub < EPW;
  const int e = (blockIdx.x * 4 + wave) * EPW + sub;
  const bool active = lane_ok && e < a.ne;
  double *sm = smem + (size_t)(wave * EPW + (lane_ok ? sub : 0)) * (2 * n1 * n1 * n1);
  const int pc = a.pc, pf = a.pf, ncc = pc + 1, nfc = pf + 1;
  if (a.kind == 1) {
    // discrete gradient (fem/bilinearform.cpp:203-282 with mfem::GradientInterpolator /
    // ProjectGrad, basis.cpp:139-143): ND dof (C; i,j,k) = d/dx_C of the H1 interpolant at the ND
    // node, i.e. the 1-D matrix Io = l_a'(open point i) along C and the identity (Ic) elsewhere
    const int Pc = nfc * nfc * nfc, Pf = 3 * pf * nfc * nfc;
    for (int C = 0; C < 3; C++) {
      const int nf0 = C == 0 ? pf : nfc, nf1 = C == 1 ? pf : nfc, nf2 = C == 2 ? pf : nfc;
      interp_block<TRANSPOSE>(a, e, active, lane_ok, ta, tb, sm, C > 0, 0, C * pf * nfc * nfc, Pc, Pf, nfc, nfc, nfc, nf0,
                              nf1, nf2, C == 0 ? a.Io : a.Ic, C == 1 ? a.Io : a.Ic, C == 2 ? a.Io : a.Ic);
    }
  } else if (a.fe_type == PA_FE_HCURL) {
    const int Pc = 3 * pc * ncc * ncc, Pf = 3 * pf * nfc * nfc;
    for (int C = 0; C < 3; C++) {
      const int nc0 = C == 0 ? pc : ncc, nc1 = C == 1 ? pc : ncc, nc2 = C == 2 ? pc : ncc;
      const int nf0 = C == 0 ? pf : nfc, nf1 = C == 1 ? pf : nfc, nf2 = C == 2 ? pf : nfc;
      interp_block<TRANSPOSE>(a, e, active, lane_ok, ta, tb, sm, false, C * pc * ncc * ncc, C * pf * nfc * nfc, Pc, Pf,
                              nc0, nc1, nc2, nf0, nf1, nf2, C == 0 ? a.Io : a.Ic, C == 1 ? a.Io : a.Ic,
                              C == 2 ? a.Io : a.Ic);
    }
  } else {
    interp_block<TRANSPOSE>(a, e, active, lane_ok, ta, tb, sm, false, 0, 0, ncc * ncc * ncc, nfc * nfc * nfc, ncc, ncc,
                            ncc, nfc, nfc, nfc, a.Ic, a.Ic, a.Ic);
  }
}

// ---- specialised forms (orders known at compile time, pf <= 4): loops unrolled, line data in registers,
// 1-D matrices as scalar operands.  Same passes and the same owner-copy semantics as above.
//
// Round 6: the memory side of the three component blocks is issued up front.  The first form of this kernel ran each block as
// index load -> x load -> three passes -> fine index load -> store, one block after the other, with 24-30 registers: four
// dependent global round trips per block and at most a handful of loads in flight per wave (p2 -> p3 prolongation 122 us for
// ~210 MB of necessary traffic).  Now: (A) every index word of the element -- coarse and fine, all blocks -- is requested first,
// (B) then every input value, (C) then the passes run block by block out of registers and (D) the stores go out.  The discrete
// gradient reads its H1 line once for the three blocks and its transpose adds the three blocks in registers before one store
// (the first form re-read and re-wrote the coarse E-vector entries twice).
template <int NC0_, int NC1_, int NC2_, int NF0_, int NF1_, int NF2_>
struct BlkDims {
  static constexpr int NC0 = NC0_, NC1 = NC1_, NC2 = NC2_, NF0 = NF0_, NF1 = NF1_, NF2 = NF2_;
};

// (A) forward: signed coarse index words of the lane's line (i = 0 .. NC0) and the fine index words of its output line
template <class D>
__device__ __forceinline__ void blk_idx_fwd(const InterpArgs &a, const int e, const bool active, const int ta, const int tb, const int off_c,
                                            const int off_f, const int Pc, const int Pf, int (&sc)[kMaxN], int (&sf)[kMaxN],
                                            const bool want_c) {
  const bool actc = ta < D::NC1 && tb < D::NC2, actf = ta < D::NF0 && tb < D::NF1;
#pragma unroll
  for (int i = 0; i < D::NC0; i++)
    sc[i] = (want_c && active && actc) ? a.lidx_c[(size_t)e * Pc + off_c + i + D::NC0 * (ta + D::NC1 * tb)] : 0;
#pragma unroll
  for (int fk = 0; fk < D::NF2; fk++)
    sf[fk] = (active && actf) ? a.lidx_f[(size_t)e * Pf + off_f + ta + D::NF0 * (tb + D::NF1 * fk)] : 0;
}
// (B) forward: the coarse values
template <class D>
__device__ __forceinline__ void blk_x_fwd(const InterpArgs &a, const bool active, const int ta, const int tb, const int (&sc)[kMaxN],
                                          double (&u)[kMaxN]) {
  const bool act = active && ta < D::NC1 && tb < D::NC2;
#pragma unroll
  for (int i = 0; i < D::NC0; i++) {
    const int s = sc[i];
    const double xv = act ? a.x[s >= 0 ? s : -1 - s] : 0.0;
    u[i] = s >= 0 ? xv : -xv;
  }
}
// (C, D) forward: passes X, Y, Z and the owner stores
template <class D, int N1>
__device__ __forceinline__ void blk_apply_fwd(const InterpArgs &a, const bool active, const bool lane_ok, const int ta, const int tb,
                                              double *sm, const double (&uin)[kMaxN], const int (&sf)[kMaxN], const double *M0,
                                              const double *M1, const double *M2) {
  constexpr int NC0 = D::NC0, NC1 = D::NC1, NC2 = D::NC2, NF0 = D::NF0, NF1 = D::NF1, NF2 = D::NF2;
  double *sA = sm, *sB = sm + N1 * N1 * N1;
  {
    const bool act = ta < NC1 && tb < NC2;
#pragma unroll
    for (int fi = 0; fi < NF0; fi++) {
      double v = 0.0;
#pragma unroll
      for (int i = 0; i < NC0; i++) v += M0[fi * NC0 + i] * uin[i];
      if (lane_ok && act) sA[(fi * NC1 + ta) * NC2 + tb] = v;
    }
  }
  wsync();
  {
    const bool act = ta < NF0 && tb < NC2;
    double u[NC1];
#pragma unroll
    for (int j = 0; j < NC1; j++) u[j] = sA[((act ? ta : 0) * NC1 + j) * NC2 + (act ? tb : 0)];
#pragma unroll
    for (int fj = 0; fj < NF1; fj++) {
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < NC1; j++) v += M1[fj * NC1 + j] * u[j];
      if (lane_ok && act) sB[(ta * NF1 + fj) * NC2 + tb] = v;
    }
  }
  wsync();
  {
    const bool act = ta < NF0 && tb < NF1;
    double u[NC2];
#pragma unroll
    for (int k = 0; k < NC2; k++) u[k] = sB[((act ? ta : 0) * NF1 + (act ? tb : 0)) * NC2 + k];
#pragma unroll
    for (int fk = 0; fk < NF2; fk++) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NC2; k++) v += M2[fk * NC2 + k] * u[k];
      if (active && act) {
        const int s = sf[fk];
        const int g = s >= 0 ? s : -1 - s;
        if (g & kOwnBit) a.y[g & ~kOwnBit] = s >= 0 ? v : -v;
      }
    }
  }
  wsync();
}
// (A) transpose: the fine index words of the lane's line; (B) the owner-masked fine values
template <class D>
__device__ __forceinline__ void blk_idx_tr(const InterpArgs &a, const int e, const bool active, const int ta, const int tb, const int off_f,
                                           const int Pf, int (&sf)[kMaxN]) {
  const bool act = active && ta < D::NF0 && tb < D::NF1;
#pragma unroll
  for (int fk = 0; fk < D::NF2; fk++) sf[fk] = act ? a.lidx_f[(size_t)e * Pf + off_f + ta + D::NF0 * (tb + D::NF1 * fk)] : 0;
}
template <class D>
__device__ __forceinline__ void blk_x_tr(const InterpArgs &a, const bool active, const int ta, const int tb, const int (&sf)[kMaxN],
                                         double (&u)[kMaxN]) {
  const bool act = active && ta < D::NF0 && tb < D::NF1;
#pragma unroll
  for (int fk = 0; fk < D::NF2; fk++) {
    const int s = sf[fk];
    const int g = s >= 0 ? s : -1 - s;
    const double xv = (act && (g & kOwnBit)) ? a.x[g & ~kOwnBit] : 0.0;
    u[fk] = s >= 0 ? xv : -xv;
  }
}
// (C) transpose: passes Z^T, Y^T, X^T; the lane's coarse line is returned in out[0 .. NC0) (added to it when `accumulate`)
template <class D, int N1>
__device__ __forceinline__ void blk_apply_tr(const bool lane_ok, const int ta, const int tb, double *sm, const double (&uin)[kMaxN],
                                             double (&out)[kMaxN], const bool accumulate, const double *M0, const double *M1,
                                             const double *M2) {
  constexpr int NC0 = D::NC0, NC1 = D::NC1, NC2 = D::NC2, NF0 = D::NF0, NF1 = D::NF1, NF2 = D::NF2;
  double *sA = sm, *sB = sm + N1 * N1 * N1;
  {
    const bool act = ta < NF0 && tb < NF1;
#pragma unroll
    for (int k = 0; k < NC2; k++) {
      double v = 0.0;
#pragma unroll
      for (int fk = 0; fk < NF2; fk++) v += M2[fk * NC2 + k] * uin[fk];
      if (lane_ok && act) sB[(ta * NF1 + tb) * NC2 + k] = v;
    }
  }
  wsync();
  {
    const bool act = ta < NF0 && tb < NC2;
    double u[NF1];
#pragma unroll
    for (int fj = 0; fj < NF1; fj++) u[fj] = sB[((act ? ta : 0) * NF1 + fj) * NC2 + (act ? tb : 0)];
#pragma unroll
    for (int j = 0; j < NC1; j++) {
      double v = 0.0;
#pragma unroll
      for (int fj = 0; fj < NF1; fj++) v += M1[fj * NC1 + j] * u[fj];
      if (lane_ok && act) sA[(ta * NC1 + j) * NC2 + tb] = v;
    }
  }
  wsync();
  {
    const bool act = ta < NC1 && tb < NC2;
    double u[NF0];
#pragma unroll
    for (int fi = 0; fi < NF0; fi++) u[fi] = sA[(fi * NC1 + (act ? ta : 0)) * NC2 + (act ? tb : 0)];
#pragma unroll
    for (int i = 0; i < NC0; i++) {
      double v = 0.0;
#pragma unroll
      for (int fi = 0; fi < NF0; fi++) v += M0[fi * NC0 + i] * u[fi];
      out[i] = accumulate ? out[i] + v : v;
    }
  }
  wsync();
}
template <class D>
__device__ __forceinline__ void blk_store_tr(const InterpArgs &a, const int e, const bool active, const int ta, const int tb, const int off_c,
                                             const int Pc, const double (&out)[kMaxN]) {
  if (!(active && ta < D::NC1 && tb < D::NC2)) return;
#pragma unroll
  for (int i = 0; i < D::NC0; i++) a.ye_c[(size_t)e * Pc + off_c + i + D::NC0 * (ta + D::NC1 * tb)] = out[i];
}

// KIND 0: ND prolongation, 1: discrete gradient H1(PF) -> ND(PF), 2: H1 prolongation
template <bool TRANSPOSE, int KIND, int PC, int PF>
__global__ __launch_bounds__(256) void interp_kernel_s(const InterpArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int N1 = PF + 1, T = N1 * N1, EPW = 64 / T, NCC = PC + 1, NFC = PF + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / T, t = lane - sub * T;
  const int ta = t % N1, tb = t / N1;
  const bool lane_ok = sub < EPW;
  const int e = (blockIdx.x * 4 + wave) * EPW + sub;
  const bool active = lane_ok && e < a.ne;
  double *sm = smem + (size_t)(wave * EPW + (lane_ok ? sub : 0)) * (2 * N1 * N1 * N1);
  const double *Ic = a.Ic_s, *Io = a.Io_s;
  if (KIND == 2) {
    using D = BlkDims<NCC, NCC, NCC, NFC, NFC, NFC>;
    constexpr int Pc = NCC * NCC * NCC, Pf = NFC * NFC * NFC;
    int sc[kMaxN], sf[kMaxN];
    double u[kMaxN], o[kMaxN];
    if (!TRANSPOSE) {
      blk_idx_fwd<D>(a, e, active, ta, tb, 0, 0, Pc, Pf, sc, sf, true);
      blk_x_fwd<D>(a, active, ta, tb, sc, u);
      blk_apply_fwd<D, N1>(a, active, lane_ok, ta, tb, sm, u, sf, Ic, Ic, Ic);
    } else {
      blk_idx_tr<D>(a, e, active, ta, tb, 0, Pf, sf);
      blk_x_tr<D>(a, active, ta, tb, sf, u);
      blk_apply_tr<D, N1>(lane_ok, ta, tb, sm, u, o, false, Ic, Ic, Ic);
      blk_store_tr<D>(a, e, active, ta, tb, 0, Pc, o);
    }
    return;
  }
  // three component blocks: the open direction of block C is C
  using D0 = typename std::conditional<KIND == 1, BlkDims<NFC, NFC, NFC, PF, NFC, NFC>, BlkDims<PC, NCC, NCC, PF, NFC, NFC>>::type;
  using D1 = typename std::conditional<KIND == 1, BlkDims<NFC, NFC, NFC, NFC, PF, NFC>, BlkDims<NCC, PC, NCC, NFC, PF, NFC>>::type;
  using D2 = typename std::conditional<KIND == 1, BlkDims<NFC, NFC, NFC, NFC, NFC, PF>, BlkDims<NCC, NCC, PC, NFC, NFC, PF>>::type;
  constexpr int Pc = KIND == 1 ? NFC * NFC * NFC : 3 * PC * NCC * NCC, Pf = 3 * PF * NFC * NFC;
  constexpr int bc = KIND == 1 ? 0 : PC * NCC * NCC, bf = PF * NFC * NFC;  // block strides in the coarse / fine element vectors
  int sf0[kMaxN], sf1[kMaxN], sf2[kMaxN];
  double u0[kMaxN], u1[kMaxN], u2[kMaxN];
  if (!TRANSPOSE) {
    int sc0[kMaxN], sc1[kMaxN], sc2[kMaxN];
    blk_idx_fwd<D0>(a, e, active, ta, tb, 0, 0, Pc, Pf, sc0, sf0, true);
    blk_idx_fwd<D1>(a, e, active, ta, tb, bc, bf, Pc, Pf, sc1, sf1, KIND != 1);
    blk_idx_fwd<D2>(a, e, active, ta, tb, 2 * bc, 2 * bf, Pc, Pf, sc2, sf2, KIND != 1);
    blk_x_fwd<D0>(a, active, ta, tb, sc0, u0);
    if (KIND != 1) {
      blk_x_fwd<D1>(a, active, ta, tb, sc1, u1);
      blk_x_fwd<D2>(a, active, ta, tb, sc2, u2);
    }
    // (the gradient's three blocks contract the same H1 line: u0)
    blk_apply_fwd<D0, N1>(a, active, lane_ok, ta, tb, sm, u0, sf0, Io, Ic, Ic);
    blk_apply_fwd<D1, N1>(a, active, lane_ok, ta, tb, sm, KIND == 1 ? u0 : u1, sf1, Ic, Io, Ic);
    blk_apply_fwd<D2, N1>(a, active, lane_ok, ta, tb, sm, KIND == 1 ? u0 : u2, sf2, Ic, Ic, Io);
  } else {
    blk_idx_tr<D0>(a, e, active, ta, tb, 0, Pf, sf0);
    blk_idx_tr<D1>(a, e, active, ta, tb, bf, Pf, sf1);
    blk_idx_tr<D2>(a, e, active, ta, tb, 2 * bf, Pf, sf2);
    blk_x_tr<D0>(a, active, ta, tb, sf0, u0);
    blk_x_tr<D1>(a, active, ta, tb, sf1, u1);
    blk_x_tr<D2>(a, active, ta, tb, sf2, u2);
    double o[kMaxN];
    blk_apply_tr<D0, N1>(lane_ok, ta, tb, sm, u0, o, false, Io, Ic, Ic);
    if (KIND != 1) blk_store_tr<D0>(a, e, active, ta, tb, 0, Pc, o);
    blk_apply_tr<D1, N1>(lane_ok, ta, tb, sm, u1, o, KIND == 1, Ic, Io, Ic);
    if (KIND != 1) blk_store_tr<D1>(a, e, active, ta, tb, bc, Pc, o);
    blk_apply_tr<D2, N1>(lane_ok, ta, tb, sm, u2, o, KIND == 1, Ic, Ic, Io);
    blk_store_tr<D2>(a, e, active, ta, tb, 2 * bc, Pc, o);  // (gradient: the sum of the three blocks, stored once)
  }
}

// coarse E^T as a gather (same scheme as pa::et_gather_kernel)
// (round 6: four dofs per thread, a block width apart, and the first four copies of each requested side by side -- one dof per
// thread with the chain tptr -> tent -> ye walked copy by copy left the kernel waiting on one load at a time: 43 us for the 3M
// coarse dofs of the p3 -> p2 restriction.  Same copies in the same order: same bits.)
// (small vectors -- the coarse levels of small problems, where the launch itself is the cost -- keep one dof per thread: more
// blocks in flight)
template <int kGatherDofs>
__global__ __launch_bounds__(256) void k_gather_t(const int n, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tent,
                                                  const double *__restrict__ ye, double *__restrict__ y) {
  const int d0 = blockIdx.x * (256 * kGatherDofs) + threadIdx.x;
  int b[kGatherDofs], e[kGatherDofs];
  double s[kGatherDofs];
  int longest = 0;
#pragma unroll
  for (int u = 0; u < kGatherDofs; u++) {
    const int d = d0 + 256 * u;
    b[u] = d < n ? tptr[d] : 0, e[u] = d < n ? tptr[d + 1] : 0;
    s[u] = 0.0;
    longest = max(longest, e[u] - b[u]);
  }
  // four copies of each dof at a time, side by side; a dof's copies are added in their order (an absent copy adds an exact zero)
  for (int q0 = 0; q0 < longest; q0 += 4) {
    int t[kGatherDofs][4];
    double v[kGatherDofs][4];
#pragma unroll
    for (int u = 0; u < kGatherDofs; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) t[u][q] = b[u] + q0 + q < e[u] ? tent[b[u] + q0 + q] : 0;
#pragma unroll
    for (int u = 0; u < kGatherDofs; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int tt = t[u][q];
        const double w = b[u] + q0 + q < e[u] ? ye[tt >= 0 ? tt : -1 - tt] : 0.0;
        v[u][q] = tt >= 0 ? w : -w;
      }
#pragma unroll
    for (int u = 0; u < kGatherDofs; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) s[u] += v[u][q];
  }
#pragma unroll
  for (int u = 0; u < kGatherDofs; u++) {
    const int d = d0 + 256 * u;
    if (d < n) y[d] = s[u];
  }
}
// G lanes per dof (G = 2, 4, 8), each adding its share of the copies, a butterfly over the group: for rows of very different lengths
// (tetrahedra: a vertex dof of an H1 space has ~24 copies, an edge dof ~5 -- a wave of the forms above runs as long as its longest
// row).  Deterministic; not the bits of the serial order.  (pa_dense.hip: et_gather_group_kernel is the same idea.)
template <int G>
__global__ __launch_bounds__(256) void k_gather_group(const int n, const int32_t *__restrict__ tptr, const int32_t *__restrict__ tent,
                                                      const double *__restrict__ ye, double *__restrict__ y) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const int d = (int)(tid / G), g = (int)(tid % G);
  const bool live = d < n;
  const int b = live ? tptr[d] : 0, e = live ? tptr[d + 1] : 0;
  double s = 0.0;
  for (int k = b + g; k < e; k += 2 * G) {
    const int t0 = tent[k], t1 = k + G < e ? tent[k + G] : 0;
    const double v0 = ye[t0 >= 0 ? t0 : -1 - t0];
    const double v1 = k + G < e ? ye[t1 >= 0 ? t1 : -1 - t1] : 0.0;
    s += t0 >= 0 ? v0 : -v0;
    s += t1 >= 0 ? v1 : -v1;
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (live && g == 0) y[d] = s;
}
// group = lanes per dof (1: the forms above); callers with irregular rows pass the power of two near their average row length
void launch_k_gather(const int n, const int32_t *tptr, const int32_t *tent, const double *ye, double *y, hipStream_t s, const int group = 1) {
  if (group > 1) {
    const unsigned nb = (unsigned)(((long long)n * group + 255) / 256);
    if (group >= 8) hipLaunchKernelGGL(k_gather_group<8>, dim3(nb), dim3(256), 0, s, n, tptr, tent, ye, y);
    else if (group >= 4) hipLaunchKernelGGL(k_gather_group<4>, dim3(nb), dim3(256), 0, s, n, tptr, tent, ye, y);
    else hipLaunchKernelGGL(k_gather_group<2>, dim3(nb), dim3(256), 0, s, n, tptr, tent, ye, y);
    return;
  }
  if (n >= (1 << 18))
    hipLaunchKernelGGL(k_gather_t<4>, dim3((n + 1023) / 1024), dim3(256), 0, s, n, tptr, tent, ye, y);
  else
    hipLaunchKernelGGL(k_gather_t<1>, dim3((n + 255) / 256), dim3(256), 0, s, n, tptr, tent, ye, y);
}

std::vector<int32_t> signed_lex_index(const pa_restriction_desc &r, const pa_basis_desc &b, int P) {
  std::vector<int32_t> lidx((size_t)r.num_elem * P);
  for (int e = 0; e < r.num_elem; e++)
    for (int l = 0; l < P; l++) {
      int n = b.dof_map ? b.dof_map[l] : l;
      bool neg = false;
      if (n < 0) n = -1 - n, neg = true;
      const size_t k = (size_t)e * P + n;
      if (r.orients && r.orients[k]) neg = !neg;
      lidx[(size_t)e * P + l] = neg ? -1 - r.offsets[k] : r.offsets[k];
    }
  return lidx;
}

// ---- dense element interpolator (tetrahedra and other non-tensor elements) ----------------------
// The libCEED interpolator operator of DiscreteLinearOperator::PartialAssemble for a general element:
// identity QFunction + the P_range x P_domain element projection matrix as "basis"
// (fem/libceed/basis.cpp:116-165), domain restriction as for any operator, range restriction built
// with the DUAL inverse transformation when the space has one (restriction.cpp:318-336): with
// domain rows T (u_e = T x_e) and range rows B (E_range^T applies B^T),
//     y_range = D^-1 sum_e  S_e^T  B_e^T  M  T_e  x_domain[off_e]
// Every copy of a shared range dof is equal, so one owner copy is stored instead of summing and
// scaling by the inverse multiplicity (as in interp_kernel above); the transpose reads through the
// same owner mask.  One wave per element; P <= 256 on either side.
struct DenseInterpArgs {
  int ne, Pd, Pr;
  const int32_t *off_d, *off_r;   // [ne][P]; range offsets carry kOwnBit on the owner copy
  const int8_t *sgn_d, *sgn_r;    // oriented: +-1 per entry, or nullptr
  const int8_t *T_d, *B_r;        // curl-oriented: [ne][P][3] {sub, main, super}, or nullptr
  const double *M;                // [nmat][Pr][Pd]
  const uint8_t *mat_id;          // [ne] which matrix an element uses (refinement transfers: the child's place in its parent), or nullptr
  const double *x;
  double *y;     // forward: range L-vector
  double *ye_d;  // transpose: domain E-vector [ne][Pd]
};

constexpr int kDenseInterpMaxP = 256;
constexpr int kDenseInterpWaves = 4;
constexpr int kDenseInterpEPW = 4;  // elements per wave and step

// Round 6: four waves per workgroup, each walking elements with a grid stride, and the element matrices staged in LDS once per
// workgroup when they fit (lds_mats > 0: all `nmat` matrices; 16 KB for the 45 x 45 matrix of order-3 Nedelec tetrahedra).  The
// first form launched one 64-lane workgroup per element and read the matrix through the vector cache for every element: 16 KB
// of cache traffic for 0.7 KB of vector data (config 3's solver loop: 10 % of its device time in these two kernels).
// NR > 0 (one matrix, P <= NR on the contracted side and <= 64 on the other): every lane keeps ITS row (forward) or column (transpose)
// of the matrix in NR registers for all the elements its wave walks; per element only the input strip is read from LDS, one broadcast
// per entry.  (With the matrix in LDS the product read 2 x 45 x 45 doubles per order-3 Nedelec element: the LDS pipe was the bound,
// ~190 us for 280k tetrahedra either way.)  NR == 0: the matrix from LDS / memory, any size, one of `nmat` per element.
template <bool TRANSPOSE, int NR = 0>
__global__ __launch_bounds__(64 * kDenseInterpWaves) void dense_interp_kernel(const DenseInterpArgs a, const int lds_mats, const int pmax) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double mreg[NR > 0 ? NR : 1];
  if (NR > 0) {
    const int nk = TRANSPOSE ? a.Pr : a.Pd, nl = TRANSPOSE ? a.Pd : a.Pr;  // contracted / lane-owned side
#pragma unroll
    for (int k = 0; k < NR; k++)
      mreg[k] = (k < nk && lane < nl) ? (TRANSPOSE ? a.M[(size_t)k * a.Pd + lane] : a.M[(size_t)lane * a.Pd + k]) : 0.0;
  }
  // (rows of the LDS copy an odd number of doubles apart: lane j reads row j in the forward product)
  const int ms = lds_mats > 0 ? (a.Pd | 1) : a.Pd, msz = a.Pr * ms;
  double *sM = dsm;                                                  // [lds_mats][Pr][Pd | 1]
  if (lds_mats > 0) {
    for (int k = threadIdx.x; k < lds_mats * a.Pr * a.Pd; k += 64 * kDenseInterpWaves) {
      const int r = k / a.Pd, c = k - r * a.Pd;  // (r runs over all rows of all matrices)
      sM[(size_t)r * ms + c] = a.M[k];
    }
    __syncthreads();
  }
  // kDenseInterpEPW consecutive elements per wave and step, every phase over all of them before the next synchronisation: their
  // index and value loads are in flight together (one element at a time left the wave waiting on one dependent chain)
  double *strip = dsm + (size_t)lds_mats * msz + (size_t)wave * (2 * kDenseInterpEPW) * pmax;
  for (int g = blockIdx.x * kDenseInterpWaves + wave; g * kDenseInterpEPW < a.ne; g += gridDim.x * kDenseInterpWaves) {
    const int e0 = g * kDenseInterpEPW;
    if (!TRANSPOSE) {
#pragma unroll
      for (int h = 0; h < kDenseInterpEPW; h++) {
        const int e = e0 + h;
        double *s0 = strip + (2 * h) * pmax;
        if (e < a.ne)
          for (int i = lane; i < a.Pd; i += 64) {
            double v = a.x[a.off_d[(size_t)e * a.Pd + i]];
            if (a.sgn_d) v *= (double)a.sgn_d[(size_t)e * a.Pd + i];
            s0[i] = v;
          }
      }
      wsync();
      if (a.T_d) {  // u = T x_e
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          const int e = e0 + h;
          double *s0 = strip + (2 * h) * pmax, *s1 = s0 + pmax;
          if (e < a.ne) {
            const int8_t *T = a.T_d + 3 * (size_t)e * a.Pd;
            for (int i = lane; i < a.Pd; i += 64)
              s1[i] = (double)T[3 * i] * s0[max(i - 1, 0)] + (double)T[3 * i + 1] * s0[i] + (double)T[3 * i + 2] * s0[min(i + 1, a.Pd - 1)];
          }
        }
        wsync();
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          double *s0 = strip + (2 * h) * pmax, *s1 = s0 + pmax;
          if (e0 + h < a.ne)
            for (int i = lane; i < a.Pd; i += 64) s0[i] = s1[i];
        }
        wsync();
      }
      if (NR > 0) {  // v = M u: the lane's row from registers, u broadcast from the strips
        double v[kDenseInterpEPW];
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) v[h] = 0.0;
#pragma unroll
        for (int i = 0; i < NR; i++)
          if (i < a.Pd) {
#pragma unroll
            for (int h = 0; h < kDenseInterpEPW; h++) v[h] += mreg[i] * strip[(2 * h) * pmax + i];
          }
        if (lane < a.Pr) {
#pragma unroll
          for (int h = 0; h < kDenseInterpEPW; h++) strip[(2 * h + 1) * pmax + lane] = v[h];
        }
      } else {
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          const int e = e0 + h;
          if (e >= a.ne) continue;
          const double *s0 = strip + (2 * h) * pmax;
          double *s1 = strip + (2 * h + 1) * pmax;
          const int mid = a.mat_id ? a.mat_id[e] : 0;
          const double *Me = lds_mats > 0 ? sM + (size_t)mid * msz : a.M + (size_t)mid * msz;
          for (int j = lane; j < a.Pr; j += 64) {  // v = M u
            const double *row = Me + (size_t)j * ms;
            double v = 0.0;
            for (int i = 0; i < a.Pd; i++) v += row[i] * s0[i];
            s1[j] = v;
          }
        }
      }
      wsync();
#pragma unroll
      for (int h = 0; h < kDenseInterpEPW; h++) {
        const int e = e0 + h;
        if (e >= a.ne) continue;
        const double *s1 = strip + (2 * h + 1) * pmax;
        const int32_t *orr = a.off_r + (size_t)e * a.Pr;
        const int8_t *B = a.B_r ? a.B_r + 3 * (size_t)e * a.Pr : nullptr;
        for (int j = lane; j < a.Pr; j += 64) {  // w = B^T v (or the sign), owner copy stored
          const int o = orr[j];
          if (!(o & kOwnBit)) continue;
          double w;
          if (B) {
            w = (double)B[3 * j + 1] * s1[j];
            if (j > 0) w += (double)B[3 * (j - 1) + 2] * s1[j - 1];
            if (j + 1 < a.Pr) w += (double)B[3 * (j + 1)] * s1[j + 1];
          } else {
            w = a.sgn_r ? (double)a.sgn_r[(size_t)e * a.Pr + j] * s1[j] : s1[j];
          }
          a.y[o & ~kOwnBit] = w;
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < kDenseInterpEPW; h++) {
        const int e = e0 + h;
        double *s0 = strip + (2 * h) * pmax;
        if (e < a.ne)
          for (int j = lane; j < a.Pr; j += 64) {  // z = owner-masked range values
            const int o = a.off_r[(size_t)e * a.Pr + j];
            double v = (o & kOwnBit) ? a.x[o & ~kOwnBit] : 0.0;
            if (a.sgn_r) v *= (double)a.sgn_r[(size_t)e * a.Pr + j];
            s0[j] = v;
          }
      }
      wsync();
      if (a.B_r) {  // v = B z
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          const int e = e0 + h;
          double *s0 = strip + (2 * h) * pmax, *s1 = s0 + pmax;
          if (e < a.ne) {
            const int8_t *B = a.B_r + 3 * (size_t)e * a.Pr;
            for (int j = lane; j < a.Pr; j += 64)
              s1[j] = (double)B[3 * j] * s0[max(j - 1, 0)] + (double)B[3 * j + 1] * s0[j] + (double)B[3 * j + 2] * s0[min(j + 1, a.Pr - 1)];
          }
        }
        wsync();
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          double *s0 = strip + (2 * h) * pmax, *s1 = s0 + pmax;
          if (e0 + h < a.ne)
            for (int j = lane; j < a.Pr; j += 64) s0[j] = s1[j];
        }
        wsync();
      }
      if (NR > 0) {  // u = M^T v: the lane's column from registers
        double v[kDenseInterpEPW];
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) v[h] = 0.0;
#pragma unroll
        for (int j = 0; j < NR; j++)
          if (j < a.Pr) {
#pragma unroll
            for (int h = 0; h < kDenseInterpEPW; h++) v[h] += mreg[j] * strip[(2 * h) * pmax + j];
          }
        if (lane < a.Pd) {
#pragma unroll
          for (int h = 0; h < kDenseInterpEPW; h++) strip[(2 * h + 1) * pmax + lane] = v[h];
        }
      } else {
#pragma unroll
        for (int h = 0; h < kDenseInterpEPW; h++) {
          const int e = e0 + h;
          if (e >= a.ne) continue;
          const double *s0 = strip + (2 * h) * pmax;
          double *s1 = strip + (2 * h + 1) * pmax;
          const int mid = a.mat_id ? a.mat_id[e] : 0;
          const double *Me = lds_mats > 0 ? sM + (size_t)mid * msz : a.M + (size_t)mid * msz;
          for (int i = lane; i < a.Pd; i += 64) {  // u = M^T v
            double v = 0.0;
            for (int j = 0; j < a.Pr; j++) v += Me[(size_t)j * ms + i] * s0[j];
            s1[i] = v;
          }
        }
      }
      wsync();
#pragma unroll
      for (int h = 0; h < kDenseInterpEPW; h++) {
        const int e = e0 + h;
        if (e >= a.ne) continue;
        const double *s1 = strip + (2 * h + 1) * pmax;
        const int8_t *T = a.T_d ? a.T_d + 3 * (size_t)e * a.Pd : nullptr;
        for (int i = lane; i < a.Pd; i += 64) {  // w = T^T u (or the sign) into the domain E-vector
          double w;
          if (T) {
            w = (double)T[3 * i + 1] * s1[i];
            if (i > 0) w += (double)T[3 * (i - 1) + 2] * s1[i - 1];
            if (i + 1 < a.Pd) w += (double)T[3 * (i + 1)] * s1[i + 1];
          } else {
            w = a.sgn_d ? (double)a.sgn_d[(size_t)e * a.Pd + i] * s1[i] : s1[i];
          }
          a.ye_d[(size_t)e * a.Pd + i] = w;
        }
      }
    }
    wsync();  // (the wave's strips are reused by its next group)
  }
}

// ---- the same operator with NOTHING BUT REGISTERS between the gather and the store (round 6) --------------------------------------
// One matrix, both element sizes <= 64, the contracted side <= 48: the LDS form above spends a wave-step of four elements on five
// dependent stages (index -> value -> strip -> synchronise -> product -> synchronise -> store), 12 us per step on config 3's mesh
// (84 us for 117k elements, of which the memory system accounts for ~20).  Here a lane keeps its row (column) of the matrix in NR
// registers as before, the tridiagonal orientation transforms take their neighbours by lane shuffles, the product takes the input
// vector entry by entry with v_readlane (a scalar operand of the FMA: no LDS, no barrier anywhere), and the loop is software-pipelined:
// index words two steps ahead, gathered values and orientation words one step ahead.  Orientation data comes as ONE packed word
// per entry, prepared on the host: the three coefficients a lane multiplies {previous, own, next} with (rows of T / B for the input
// side, columns for the output side, zero beyond the ends; plain or sign-only restrictions are the tridiagonal (0, +-1, 0)).
struct DenseInterpRegArgs {
  int ne, Pd, Pr;
  const int32_t *off_in, *off_r;  // input side index ([ne][Pin]; transposed: the range offsets with kOwnBit), range offsets (forward store)
  const int32_t *pk_in, *pk_out;  // packed {prev, own, next} coefficients of the input / output side
  const double *M, *x;
  double *y, *ye_d;
};
__device__ __forceinline__ double readlane_f64(const double v, const int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tri3(const int pk, const double m, const double c, const double p) {
  return (double)(int8_t)(pk & 0xff) * m + (double)(int8_t)((pk >> 8) & 0xff) * c + (double)(int8_t)((pk >> 16) & 0xff) * p;
}
constexpr int kRegE = 4;  // elements per wave and step
template <bool TRANSPOSE, int NR>
__global__ __launch_bounds__(256) void dense_interp_reg_kernel(const DenseInterpRegArgs a) {
  const int lane = threadIdx.x & 63;
  const int Pin = TRANSPOSE ? a.Pr : a.Pd, Pout = TRANSPOSE ? a.Pd : a.Pr;
  double mreg[NR];
#pragma unroll
  for (int k = 0; k < NR; k++)
    mreg[k] = (k < Pin && lane < Pout) ? (TRANSPOSE ? a.M[(size_t)k * a.Pd + lane] : a.M[(size_t)lane * a.Pd + k]) : 0.0;
  const int ngroups = (a.ne + kRegE - 1) / kRegE, nw = gridDim.x * 4;
  int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= ngroups) return;
  const bool lin = lane < Pin, lout = lane < Pout;
  auto el = [&](const int gg, const int h) { return min(gg * kRegE + h, a.ne - 1); };  // (clamped: loads stay in range, stores are masked)
  auto load_idx = [&](const int gg, int (&idx)[kRegE]) {
#pragma unroll
    for (int h = 0; h < kRegE; h++) idx[h] = lin ? a.off_in[(size_t)el(gg, h) * Pin + lane] : 0;
  };
  auto load_x = [&](const int (&idx)[kRegE], double (&xv)[kRegE]) {
#pragma unroll
    for (int h = 0; h < kRegE; h++) {
      if (TRANSPOSE) xv[h] = (lin && (idx[h] & kOwnBit)) ? a.x[idx[h] & ~kOwnBit] : 0.0;  // owner-masked range values
      else xv[h] = lin ? a.x[idx[h]] : 0.0;
    }
  };
  auto load_or = [&](const int gg, int (&pin)[kRegE], int (&pout)[kRegE], int (&oo)[kRegE]) {
#pragma unroll
    for (int h = 0; h < kRegE; h++) {
      const size_t e = (size_t)el(gg, h);
      pin[h] = lin ? a.pk_in[e * Pin + lane] : 0;
      pout[h] = lout ? a.pk_out[e * Pout + lane] : 0;
      oo[h] = (!TRANSPOSE && lout) ? a.off_r[e * Pout + lane] : 0;
    }
  };
  int idx1[kRegE], idx2[kRegE], pin0[kRegE], pout0[kRegE], oo0[kRegE], pin1[kRegE], pout1[kRegE], oo1[kRegE];
  double x0[kRegE], x1[kRegE];
  // prologue: everything of the first step, the index words of the second
  load_idx(g, idx1);
  load_or(g, pin0, pout0, oo0);
  load_x(idx1, x0);
  load_idx(min(g + nw, ngroups - 1), idx1);
  for (; g < ngroups; g += nw) {
    const int g1 = min(g + nw, ngroups - 1), g2 = min(g + 2 * nw, ngroups - 1);
    load_x(idx1, x1);             // values of the next step (its index words were requested a step ago)
    load_idx(g2, idx2);           // index words two steps ahead
    load_or(g1, pin1, pout1, oo1);  // orientation words and store offsets of the next step
#pragma unroll
    for (int h = 0; h < kRegE; h++) {
      const double u = x0[h];
      // (the shuffles outside the selects: inside, the last lane in range would read a neighbour that is masked off)
      const double su = __shfl_up(u, 1, 64), sd = __shfl_down(u, 1, 64);
      const double um = lane > 0 ? su : u, up = lane + 1 < Pin ? sd : u;
      const double t = lin ? tri3(pin0[h], um, u, up) : 0.0;  // T x_e (forward) / B z (transposed)
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < NR; k++) v += mreg[k] * readlane_f64(t, k);
      const double vm = __shfl_up(v, 1, 64), vp = __shfl_down(v, 1, 64);
      const double w = tri3(pout0[h], vm, v, vp);  // B^T v (forward) / T^T u (transposed): the word is zero beyond the ends
      const int e = g * kRegE + h;
      if (e < a.ne && lout) {
        if (TRANSPOSE) a.ye_d[(size_t)e * a.Pd + lane] = w;
        else if (oo0[h] & kOwnBit) a.y[oo0[h] & ~kOwnBit] = w;
      }
    }
#pragma unroll
    for (int h = 0; h < kRegE; h++) x0[h] = x1[h], pin0[h] = pin1[h], pout0[h] = pout1[h], oo0[h] = oo1[h], idx1[h] = idx2[h];
  }
}

class DenseInterpOperator : public Operator {
  const Context *ctx_;
  const Halo *halo_d_;
  int ne_, Pd_, Pr_, nl_d_, nl_r_, nt_d_, nt_r_;
  int32_t *d_off_d_ = nullptr, *d_off_r_ = nullptr, *d_tptr_ = nullptr, *d_tent_ = nullptr;
  int8_t *d_sgn_d_ = nullptr, *d_sgn_r_ = nullptr, *d_T_d_ = nullptr, *d_B_r_ = nullptr;
  double *d_M_ = nullptr, *d_ye_ = nullptr;
  uint8_t *d_mat_id_ = nullptr;
  int32_t *d_pk_[4] = {nullptr, nullptr, nullptr, nullptr};  // register form: {domain rows, domain columns, range rows, range columns}
  mutable Vector ld_, lr_;

  int nmat_ = 1, gather_group_ = 1;
  template <bool TR>
  void launch(const double *x, double *y) const {
    if (d_pk_[0]) {  // one matrix, both sides within a wave: registers only (dense_interp_reg_kernel)
      const DenseInterpRegArgs r{ne_, Pd_, Pr_, TR ? d_off_r_ : d_off_d_, d_off_r_, TR ? d_pk_[2] : d_pk_[0], TR ? d_pk_[1] : d_pk_[3],
                                 d_M_, x, y, d_ye_};
      const int nkr = TR ? Pr_ : Pd_, ngroups = (ne_ + kRegE - 1) / kRegE;
      auto go = [&](auto kernel) {
        static int per_cu = 0, n_cu = 0;  // (per instantiation: the lambda's statics belong to its closure type)
        if (!per_cu) {
          int dev = 0;
          PA_HIP(hipGetDevice(&dev));
          PA_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        }
        const int grid = std::max(1, std::min((ngroups + 3) / 4, n_cu * per_cu));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, ctx_->stream, r);
      };
      if (nkr <= 8) go(dense_interp_reg_kernel<TR, 8>);
      else if (nkr <= 24) go(dense_interp_reg_kernel<TR, 24>);
      else go(dense_interp_reg_kernel<TR, 48>);
      PA_HIP(hipGetLastError());
      return;
    }
    DenseInterpArgs a{ne_, Pd_, Pr_, d_off_d_, d_off_r_, d_sgn_d_, d_sgn_r_, d_T_d_, d_B_r_, d_M_, d_mat_id_, x, y, d_ye_};
    const int pmax = (std::max(Pd_, Pr_) + 1) & ~1;
    // the matrices in LDS when all of them fit beside the waves' strips in 48 KB (eight such workgroups per CU)
    const size_t strips = sizeof(double) * (size_t)kDenseInterpWaves * 2 * kDenseInterpEPW * pmax, mats = sizeof(double) * (size_t)nmat_ * Pr_ * (Pd_ | 1);
    // one matrix whose contracted side fits 48 registers and whose other side one wave: rows / columns in registers
    const int nk = TR ? Pr_ : Pd_, nl = TR ? Pd_ : Pr_;
    const int nr = (nmat_ == 1 && nl <= 64 && nk <= 48) ? (nk <= 8 ? 8 : nk <= 24 ? 24 : 48) : 0;
    const int lds_mats = (nr == 0 && strips + mats <= 48 * 1024) ? nmat_ : 0;
    const size_t lds = strips + (lds_mats ? mats : 0);
    // eight elements per wave (the matrix copy is amortised over 32 elements of a workgroup), every CU busy
    const int groups = (ne_ + kDenseInterpEPW - 1) / kDenseInterpEPW;  // of elements: one per wave and step; two steps per wave at size
    const int grid = std::max(1, std::min((groups + kDenseInterpWaves - 1) / kDenseInterpWaves, std::max(2048, (groups + 2 * kDenseInterpWaves - 1) / (2 * kDenseInterpWaves))));
    if (nr == 8)
      hipLaunchKernelGGL((dense_interp_kernel<TR, 8>), dim3(grid), dim3(64 * kDenseInterpWaves), lds, ctx_->stream, a, 0, pmax);
    else if (nr == 24)
      hipLaunchKernelGGL((dense_interp_kernel<TR, 24>), dim3(grid), dim3(64 * kDenseInterpWaves), lds, ctx_->stream, a, 0, pmax);
    else if (nr == 48)
      hipLaunchKernelGGL((dense_interp_kernel<TR, 48>), dim3(grid), dim3(64 * kDenseInterpWaves), lds, ctx_->stream, a, 0, pmax);
    else
      hipLaunchKernelGGL((dense_interp_kernel<TR, 0>), dim3(grid), dim3(64 * kDenseInterpWaves), lds, ctx_->stream, a, lds_mats, pmax);
    PA_HIP(hipGetLastError());
  }
  static int8_t *signs(const pa_restriction_desc &r, hipStream_t s) {
    if (!r.orients) return nullptr;
    std::vector<int8_t> v((size_t)r.num_elem * r.elem_size);
    for (size_t k = 0; k < v.size(); k++) v[k] = r.orients[k] ? -1 : 1;
    return pa::dev_upload(v.data(), v.size(), s);
  }

public:
  // nmat > 1 with mat_id [ne]: element e uses M[mat_id[e]] -- the transfer between a mesh and its uniform refinement
  // (mfem::TransferOperator between different meshes, fem/fespace.cpp:246-251): the elements are the FINE mesh's, the domain
  // restriction lists the dofs of each one's PARENT, the matrix is the local interpolation for the child's place in its parent
  DenseInterpOperator(const Context &ctx, const pa_restriction_desc &rd, const pa_restriction_desc &rr, const double *M,
                      const Halo *halo_d, int nt_d, int nt_r, int nmat = 1, const uint8_t *mat_id = nullptr)
      : Operator(nt_r, nt_d), ctx_(&ctx), halo_d_(halo_d), ne_(rd.num_elem), Pd_(rd.elem_size), Pr_(rr.elem_size),
        nl_d_(rd.lsize), nl_r_(rr.lsize), nt_d_(nt_d), nt_r_(nt_r) {
    PA_REQUIRE(rd.num_elem == rr.num_elem, "interpolation needs the same elements on both sides");
    PA_REQUIRE(nmat >= 1 && nmat <= 256 && (nmat == 1 || mat_id), "element matrices: one, or up to 256 with an index per element");
    if (mat_id)
      for (int e = 0; e < ne_; e++) PA_REQUIRE(mat_id[e] < nmat, "element matrix index out of range");
    PA_REQUIRE(M && rd.offsets && rr.offsets, "null argument");
    PA_REQUIRE(Pd_ <= kDenseInterpMaxP && Pr_ <= kDenseInterpMaxP, "element too large for the dense interpolator");
    PA_REQUIRE(!(rd.orients && rd.curl_orients) && !(rr.orients && rr.curl_orients), "restriction is oriented or curl-oriented");
    PA_REQUIRE(nt_d <= nl_d_ && nt_r <= nl_r_, "true dof counts exceed local sizes");
    PA_REQUIRE(halo_d || nt_d == nl_d_, "ghost dofs on the domain side need a halo plan");
    PA_REQUIRE(nl_r_ < kOwnBit, "too many range dofs for the owner-flag encoding");
    const size_t nd = (size_t)ne_ * Pd_, nr = (size_t)ne_ * Pr_;
    std::vector<int32_t> offr(rr.offsets, rr.offsets + nr);
    {
      std::vector<char> seen((size_t)nl_r_, 0);
      for (auto &o : offr) {
        PA_REQUIRE(o >= 0 && o < nl_r_, "range offset out of range");
        if (!seen[o]) seen[o] = 1, o |= kOwnBit;
      }
    }
    {
      std::vector<int32_t> tptr((size_t)nl_d_ + 1, 0), tent(nd);
      for (size_t k = 0; k < nd; k++) {
        PA_REQUIRE(rd.offsets[k] >= 0 && rd.offsets[k] < nl_d_, "domain offset out of range");
        tptr[(size_t)rd.offsets[k] + 1]++;
      }
      for (int d = 0; d < nl_d_; d++) tptr[d + 1] += tptr[d];
      std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
      for (size_t k = 0; k < nd; k++) tent[fill[rd.offsets[k]]++] = (int32_t)k;
      d_tptr_ = pa::dev_upload(tptr.data(), tptr.size(), ctx.stream);
      d_tent_ = pa::dev_upload(tent.data(), tent.size(), ctx.stream);
      d_ye_ = pa::dev_alloc<double>(nd);
      // rows of very different lengths (simplices): several lanes per dof in the transposed gather (PALACE_AMD_INTERP_GATHER_GROUP=0: off)
      int longest = 0;
      for (int d = 0; d < nl_d_; d++) longest = std::max(longest, tptr[d + 1] - tptr[d]);
      const double avg = nl_d_ > 0 ? (double)nd / nl_d_ : 1.0;
      const char *ge = std::getenv("PALACE_AMD_INTERP_GATHER_GROUP");
      if (longest >= 12 && !(ge && ge[0] == '0')) gather_group_ = avg < 2.5 ? 2 : avg < 5.0 ? 4 : 8;
    }
    d_off_d_ = pa::dev_upload(rd.offsets, nd, ctx.stream);
    d_off_r_ = pa::dev_upload(offr.data(), nr, ctx.stream);
    d_sgn_d_ = signs(rd, ctx.stream), d_sgn_r_ = signs(rr, ctx.stream);
    if (rd.curl_orients) d_T_d_ = pa::dev_upload(rd.curl_orients, 3 * nd, ctx.stream);
    if (rr.curl_orients) d_B_r_ = pa::dev_upload(rr.curl_orients, 3 * nr, ctx.stream);
    d_M_ = pa::dev_upload(M, (size_t)nmat * Pr_ * Pd_, ctx.stream);
    nmat_ = nmat;
    if (mat_id) d_mat_id_ = pa::dev_upload(mat_id, (size_t)ne_, ctx.stream);
    // the register form (read at creation: PALACE_AMD_DENSE_INTERP=lds keeps the strips in LDS)
    const char *form = std::getenv("PALACE_AMD_DENSE_INTERP");
    if (nmat == 1 && Pd_ <= 64 && Pr_ <= 64 && std::min(Pd_, Pr_) >= 1 && std::max(Pd_, Pr_) <= 48 && !(form && std::string(form) == "lds")) {
      auto pack = [&](const pa_restriction_desc &r, int P, bool columns) {
        std::vector<int32_t> pk((size_t)ne_ * P);
        for (int e = 0; e < ne_; e++)
          for (int i = 0; i < P; i++) {
            const size_t k = (size_t)e * P + i;
            int8_t b0 = 0, b1 = 1, b2 = 0;
            if (r.curl_orients) {
              const int8_t *t = r.curl_orients + 3 * k;
              if (columns) b0 = i > 0 ? t[-3 + 2] : 0, b1 = t[1], b2 = i + 1 < P ? t[3 + 0] : 0;  // T[i-1][i], T[i][i], T[i+1][i]
              else b0 = t[0], b1 = t[1], b2 = t[2];
            } else if (r.orients) {
              b1 = r.orients[k] ? -1 : 1;
            }
            pk[k] = (int32_t)((uint32_t)(uint8_t)b0 | ((uint32_t)(uint8_t)b1 << 8) | ((uint32_t)(uint8_t)b2 << 16));
          }
        return pa::dev_upload(pk.data(), pk.size(), ctx.stream);
      };
      d_pk_[0] = pack(rd, Pd_, false), d_pk_[1] = pack(rd, Pd_, true), d_pk_[2] = pack(rr, Pr_, false), d_pk_[3] = pack(rr, Pr_, true);
    }
    ld_.SetSize(nl_d_), lr_.SetSize(nl_r_);
  }
  ~DenseInterpOperator() override {
    (void)hipFree(d_off_d_), (void)hipFree(d_off_r_), (void)hipFree(d_tptr_), (void)hipFree(d_tent_);
    (void)hipFree(d_sgn_d_), (void)hipFree(d_sgn_r_), (void)hipFree(d_T_d_), (void)hipFree(d_B_r_);
    (void)hipFree(d_M_), (void)hipFree(d_ye_), (void)hipFree(d_mat_id_);
    for (int32_t *p : d_pk_) (void)hipFree(p);
  }
  void Mult(const Vector &x, Vector &y) const override {
    const Context &c = *ctx_;
    PA_REQUIRE(x.Size() == nt_d_ && y.Size() == nt_r_, "size mismatch in interpolation");
    if (!halo_d_ && nt_d_ == nl_d_ && nt_r_ == nl_r_) {
      launch<false>(x.Data(), y.Data());
      return;
    }
    Vector td(ld_.Data(), nt_d_);
    linalg::Copy(c, x, td);
    if (halo_d_) halo_d_->Prolongate(ld_.Data(), c.stream);
    launch<false>(ld_.Data(), lr_.Data());
    Vector tr(lr_.Data(), nt_r_);
    linalg::Copy(c, tr, y);
  }
  void MultTranspose(const Vector &x, Vector &y) const override {
    const Context &c = *ctx_;
    PA_REQUIRE(x.Size() == nt_r_ && y.Size() == nt_d_, "size mismatch in restriction");
    const bool serial = !halo_d_ && nt_d_ == nl_d_ && nt_r_ == nl_r_;
    if (!serial) {
      Vector tr(lr_.Data(), nt_r_);
      linalg::Copy(c, x, tr);
      if (nl_r_ > nt_r_)
        PA_HIP(hipMemsetAsync(lr_.Data() + nt_r_, 0, sizeof(double) * (size_t)(nl_r_ - nt_r_), c.stream));
    }
    launch<true>(serial ? x.Data() : lr_.Data(), nullptr);
    launch_k_gather(nl_d_, d_tptr_, d_tent_, d_ye_, serial ? y.Data() : ld_.Data(), c.stream, gather_group_);
    PA_HIP(hipGetLastError());
    if (serial) return;
    if (halo_d_) halo_d_->RestrictAdd(ld_.Data(), c.stream);
    Vector td(ld_.Data(), nt_d_);
    linalg::Copy(c, td, y);
  }
};

}  // namespace

Operator *make_dense_interp_operator(const Context &ctx, const pa_restriction_desc &rd, const pa_restriction_desc &rr,
                                     const double *M, const Halo *halo_d, int nt_d, int nt_r, int nmat, const uint8_t *mat_id) {
  return new DenseInterpOperator(ctx, rd, rr, M, halo_d, nt_d, nt_r, nmat, mat_id);
}

// The prolongation as an Operator on T-vectors: Mult coarse -> fine, MultTranspose fine -> coarse.
class InterpOperator : public Operator {
  const Context *ctx_;
  const Halo *halo_c_;
  int kind_ = 0;
  int fe_type_, pc_, pf_, ne_, nl_c_, nl_f_, nt_c_, nt_f_;
  int32_t *d_lidx_c_ = nullptr, *d_lidx_f_ = nullptr;
  double *d_Ic_ = nullptr, *d_Io_ = nullptr, *d_ye_c_ = nullptr;
  int32_t *d_tptr_c_ = nullptr, *d_tent_c_ = nullptr;
  std::vector<double> h_Ic_, h_Io_;
  mutable Vector lc_, lf_;

  template <bool TR>
  void launch(const double *x, double *y) const {
    InterpArgs a{kind_, ne_, fe_type_, pc_, pf_, d_lidx_c_, d_lidx_f_, d_Ic_, d_Io_, x, y, d_ye_c_, {}, {}};
    const int n1 = pf_ + 1, epw = 64 / (n1 * n1), epb = 4 * epw;
    const size_t lds = sizeof(double) * (size_t)epb * 2 * n1 * n1 * n1;
    const dim3 grid((ne_ + epb - 1) / epb), block(256);
    bool done = false;
    if (pf_ <= 4 && h_Ic_.size() <= 25 && h_Io_.size() <= 20) {
      for (size_t i = 0; i < h_Ic_.size(); i++) a.Ic_s[i] = h_Ic_[i];
      for (size_t i = 0; i < h_Io_.size(); i++) a.Io_s[i] = h_Io_[i];
      const int k = kind_ == 1 ? 1 : (fe_type_ == PA_FE_HCURL ? 0 : 2);
      done = true;
#define PA_INTERP_CASE(K, PC, PF)                                                                      \
  case (K) * 100 + (PC) * 10 + (PF):                                                                     \
    hipLaunchKernelGGL((interp_kernel_s<TR, K, PC, PF>), grid, block, lds, ctx_->stream, a);             \
    break;
      switch (k * 100 + pc_ * 10 + pf_) {
        PA_INTERP_CASE(0, 1, 2) PA_INTERP_CASE(0, 1, 3) PA_INTERP_CASE(0, 2, 3) PA_INTERP_CASE(0, 1, 4)
        PA_INTERP_CASE(0, 2, 4) PA_INTERP_CASE(0, 3, 4)
        PA_INTERP_CASE(2, 1, 2) PA_INTERP_CASE(2, 1, 3) PA_INTERP_CASE(2, 2, 3) PA_INTERP_CASE(2, 1, 4)
        PA_INTERP_CASE(2, 2, 4) PA_INTERP_CASE(2, 3, 4)
        PA_INTERP_CASE(1, 1, 1) PA_INTERP_CASE(1, 2, 2) PA_INTERP_CASE(1, 3, 3) PA_INTERP_CASE(1, 4, 4)
        default: done = false;
      }
#undef PA_INTERP_CASE
    }
    if (!done) hipLaunchKernelGGL((interp_kernel<TR>), grid, block, lds, ctx_->stream, a);
    PA_HIP(hipGetLastError());
  }

public:
  InterpOperator(const Context &ctx, const pa_restriction_desc &rc, const pa_basis_desc &bc,
                 const pa_restriction_desc &rf, const pa_basis_desc &bf, const double *Ic, const double *Io,
                 const Halo *halo_c, int nt_c, int nt_f, int kind)
      : Operator(nt_f, nt_c), ctx_(&ctx), halo_c_(halo_c), kind_(kind), fe_type_(bc.fe_type), pc_(bc.order),
        pf_(bf.order), ne_(rc.num_elem), nl_c_(rc.lsize), nl_f_(rf.lsize), nt_c_(nt_c), nt_f_(nt_f) {
    if (kind == 1) {
      PA_REQUIRE(bc.fe_type == PA_FE_H1 && bf.fe_type == PA_FE_HCURL && bc.order == bf.order,
                 "discrete gradient maps H1(p) to ND(p)");
    } else {
      PA_REQUIRE(bc.fe_type == bf.fe_type, "prolongation needs the same element family on both levels");
    }
    PA_REQUIRE(rc.num_elem == rf.num_elem, "prolongation needs the same mesh on both levels");
    PA_REQUIRE(pf_ + 1 <= kMaxN && pc_ <= pf_, "unsupported orders for prolongation");
    PA_REQUIRE(Ic && (kind == 0 && fe_type_ == PA_FE_H1 ? true : Io != nullptr), "1-D interpolation matrices missing");
    PA_REQUIRE(nt_c <= nl_c_ && nt_f <= nl_f_, "true dof counts exceed local sizes");
    PA_REQUIRE(halo_c || nt_c == nl_c_, "ghost dofs on the coarse level need a halo plan");
    const int Pc = bc.fe_type == PA_FE_HCURL ? 3 * pc_ * (pc_ + 1) * (pc_ + 1) : (pc_ + 1) * (pc_ + 1) * (pc_ + 1);
    const int Pf = bf.fe_type == PA_FE_HCURL ? 3 * pf_ * (pf_ + 1) * (pf_ + 1) : (pf_ + 1) * (pf_ + 1) * (pf_ + 1);
    PA_REQUIRE(rc.elem_size == Pc && rf.elem_size == Pf, "restriction sizes do not match the bases");
    auto lc = signed_lex_index(rc, bc, Pc), lf = signed_lex_index(rf, bf, Pf);
    PA_REQUIRE(nl_f_ < kOwnBit, "too many fine dofs for the owner-flag encoding");
    {  // owner copy of every fine dof = its first occurrence in element order
      std::vector<char> seen((size_t)nl_f_, 0);
      for (auto &sg : lf) {
        const int g = sg >= 0 ? sg : -1 - sg;
        if (!seen[g]) {
          seen[g] = 1;
          sg = sg >= 0 ? (g | kOwnBit) : -1 - (g | kOwnBit);
        }
      }
    }
    {  // transpose map of the coarse index array
      const size_t nnz = lc.size();
      std::vector<int32_t> tptr((size_t)nl_c_ + 1, 0), tent(nnz);
      for (size_t k = 0; k < nnz; k++) tptr[(size_t)(lc[k] >= 0 ? lc[k] : -1 - lc[k]) + 1]++;
      for (int d = 0; d < nl_c_; d++) tptr[d + 1] += tptr[d];
      std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
      for (size_t k = 0; k < nnz; k++) {
        const int d = lc[k] >= 0 ? lc[k] : -1 - lc[k];
        tent[fill[d]++] = lc[k] >= 0 ? (int32_t)k : -1 - (int32_t)k;
      }
      d_tptr_c_ = pa::dev_upload(tptr.data(), tptr.size(), ctx.stream);
      d_tent_c_ = pa::dev_upload(tent.data(), tent.size(), ctx.stream);
      d_ye_c_ = pa::dev_alloc<double>(nnz);
    }
    d_lidx_c_ = pa::dev_upload(lc.data(), lc.size(), ctx.stream);
    d_lidx_f_ = pa::dev_upload(lf.data(), lf.size(), ctx.stream);
    d_Ic_ = pa::dev_upload(Ic, (size_t)(pf_ + 1) * (pc_ + 1), ctx.stream);
    h_Ic_.assign(Ic, Ic + (size_t)(pf_ + 1) * (pc_ + 1));
    if (Io) {
      const size_t nio = kind == 1 ? (size_t)pf_ * (pf_ + 1) : (size_t)pf_ * pc_;
      d_Io_ = pa::dev_upload(Io, nio, ctx.stream);
      h_Io_.assign(Io, Io + nio);
    }
    lc_.SetSize(nl_c_), lf_.SetSize(nl_f_);
  }
  ~InterpOperator() override {
    (void)hipFree(d_lidx_c_), (void)hipFree(d_lidx_f_), (void)hipFree(d_Ic_), (void)hipFree(d_Io_),
        (void)hipFree(d_ye_c_), (void)hipFree(d_tptr_c_), (void)hipFree(d_tent_c_);
  }
  // y_f = R_f D^-1 E_f^T I E_c P_c x_c
  void Mult(const Vector &x, Vector &y) const override {
    const Context &c = *ctx_;
    PA_REQUIRE(x.Size() == nt_c_ && y.Size() == nt_f_, "size mismatch in prolongation");
    if (!halo_c_ && nt_c_ == nl_c_ && nt_f_ == nl_f_) {  // one rank: no staging copies
      launch<false>(x.Data(), y.Data());
      return;
    }
    Vector tc(lc_.Data(), nt_c_);
    linalg::Copy(c, x, tc);
    if (halo_c_) halo_c_->Prolongate(lc_.Data(), c.stream);
    launch<false>(lc_.Data(), lf_.Data());  // every fine local dof is stored exactly once
    Vector tf(lf_.Data(), nt_f_);
    linalg::Copy(c, tf, y);
  }
  // x_c = P_c^T E_c^T I^T E_f D^-1 R_f^T y_f
  void MultTranspose(const Vector &x, Vector &y) const override {
    const Context &c = *ctx_;
    PA_REQUIRE(x.Size() == nt_f_ && y.Size() == nt_c_, "size mismatch in restriction");
    if (!halo_c_ && nt_c_ == nl_c_ && nt_f_ == nl_f_) {
      launch<true>(x.Data(), nullptr);
      launch_k_gather(nl_c_, d_tptr_c_, d_tent_c_, d_ye_c_, y.Data(), c.stream);
      PA_HIP(hipGetLastError());
      return;
    }
    Vector tf(lf_.Data(), nt_f_);
    linalg::Copy(c, x, tf);
    if (nl_f_ > nt_f_)
      PA_HIP(hipMemsetAsync(lf_.Data() + nt_f_, 0, sizeof(double) * (size_t)(nl_f_ - nt_f_), c.stream));
    launch<true>(lf_.Data(), nullptr);
    launch_k_gather(nl_c_, d_tptr_c_, d_tent_c_, d_ye_c_, lc_.Data(), c.stream);
    PA_HIP(hipGetLastError());
    if (halo_c_) halo_c_->RestrictAdd(lc_.Data(), c.stream);
    Vector tc(lc_.Data(), nt_c_);
    linalg::Copy(c, tc, y);
  }
};

Operator *make_interp_operator(const Context &ctx, const pa_restriction_desc &rc, const pa_basis_desc &bc,
                               const pa_restriction_desc &rf, const pa_basis_desc &bf, const double *Ic,
                               const double *Io, const Halo *halo_c, int nt_c, int nt_f, int kind) {
  return new InterpOperator(ctx, rc, bc, rf, bf, Ic, Io, halo_c, nt_c, nt_f, kind);
}

}  // namespace palace
