// Shared device code of the H(curl) hexahedron kernels (pa_nd_hex.hip, pa_nd_hex_stream.hip): the mirror-symmetric
// half tables, the LDS layouts of the contraction buffers and the sum-factorised forward / transposed passes of one
// vector component.  See pa_nd_hex.hip for the mapping of elements and lines to lanes.
#pragma once

#include <type_traits>

#include "pa_internal.hpp"
#include "pa_device.hpp"

namespace pa {

// The 1-D tables are mirror-symmetric (Gauss-Legendre / Gauss-Lobatto nodes and points):
//   B[q][i] = B[Q1-1-q][n-1-i],  G[q][i] = -G[Q1-1-q][n-1-i].
// Only the first (Q1+1)/2 rows travel as kernel arguments, so every entry stays in an SGPR for the
// whole kernel (p = 3: 22 doubles instead of 44) and nothing is spilled.  pa_op_add_sub verifies
// the symmetry of the tables it is given.
// For odd Q1 the middle row is its own mirror image, so only its first half is kept as well
// (p = 4: 35 doubles instead of 42 -- the Q1 = 5 instantiations are the ones short of SGPRs).
template <int N, int Q1>
struct HalfTab {
  static constexpr int QH = (Q1 + 1) / 2;
  static constexpr int LEN = (Q1 & 1) ? (QH - 1) * N + (N + 1) / 2 : QH * N;
};
template <int P1, int Q1>
struct NDTab {
  static constexpr int QH = (Q1 + 1) / 2;
  double Bo[HalfTab<P1, Q1>::LEN];
  double Bc[HalfTab<P1 + 1, Q1>::LEN];
  double Gc[HalfTab<P1 + 1, Q1>::LEN];
};

// value-type (even symmetry) and derivative-type (odd symmetry) table access; q and i are
// compile-time constants after unrolling
template <int N, int Q1>
__device__ __forceinline__ double tab_even(const double *H, const int q, const int i) {
  constexpr int QH = (Q1 + 1) / 2;
  const int qq = (q < QH) ? q : Q1 - 1 - q, ii = (q < QH) ? i : N - 1 - i;
  if ((Q1 & 1) && qq == QH - 1) return H[(QH - 1) * N + (ii < (N + 1) / 2 ? ii : N - 1 - ii)];
  return H[qq * N + ii];
}
template <int N, int Q1>
__device__ __forceinline__ double tab_odd(const double *H, const int q, const int i) {
  constexpr int QH = (Q1 + 1) / 2;
  const int qq = (q < QH) ? q : Q1 - 1 - q, ii = (q < QH) ? i : N - 1 - i;
  const double sg = (q < QH) ? 1.0 : -1.0;
  if ((Q1 & 1) && qq == QH - 1) {  // middle row: antisymmetric in i, centre entry zero
    if ((N & 1) && ii == N / 2) return 0.0;
    return (ii < N / 2) ? sg * H[(QH - 1) * N + ii] : -sg * H[(QH - 1) * N + (N - 1 - ii)];
  }
  return sg * H[qq * N + ii];
}

// LDS strides of the two contraction buffers, A[f][qx][j][k] and B[f][qx][qy][k].  A dense layout
// makes three of the four lane patterns 4-way bank conflicted at p = 3 (ds_read_b64: 32-lane groups
// over 64 dword banks; ds_write_b64: 16-lane groups over 32); the padded strides below come from
// scripts/lds_layout_search.py and are conflict-free for the listed (P1, Q1).
template <int P1, int Q1>
struct NDStrides {
  static constexpr int NC = P1 + 1;
  static constexpr int Sj = NC, Sq = NC * NC, Ty = NC, Tq = NC * Q1, EPAD = 0;
};
template <>
struct NDStrides<2, 4> {
  static constexpr int Sj = 3, Sq = 12, Ty = 3, Tq = 12, EPAD = 0;  // 240
};
template <>
struct NDStrides<1, 4> {
  static constexpr int Sj = 2, Sq = 4, Ty = 3, Tq = 12, EPAD = 0;  // 176
};

template <int P1, int Q1>
struct NDLayout {
  using S = NDStrides<P1, Q1>;
  static constexpr int NC = P1 + 1;
  static constexpr int T = Q1 * Q1;
  static constexpr int EPW = 64 / T;
  // LDS per element: A = 2 fields [Q1][NC][NC] (after pass X), B = 3 fields [Q1][Q1][NC]
  static constexpr int A_FIELD = S::Sq * Q1;
  static constexpr int B_FIELD = S::Tq * Q1;
  static constexpr int ELEM = 2 * A_FIELD + 3 * B_FIELD;
  static constexpr bool TUNED = (S::Sq != NC * NC) || (S::Tq != NC * Q1) || (P1 == 1 && Q1 == 4);
  // tuned layouts carry their own element stride; otherwise an odd multiple of 16 doubles, so the
  // two elements of a 32-lane read group land on opposite halves of the 64 banks
  static constexpr int ELEM_PAD = TUNED ? ELEM + S::EPAD : (((ELEM + 15) / 16 * 16) | 16);
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) {
    return f * A_FIELD + qx * S::Sq + j * S::Sj + k;
  }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) {
    return 2 * A_FIELD + f * B_FIELD + qx * S::Tq + qy * S::Ty + k;
  }
  __device__ static __forceinline__ int parity_xor(int) { return 0; }
};

// p = 3 (NC = Q1 = 4): every buffer is a 4x4x4 block, so an XOR swizzle makes all four lane patterns
// conflict-free without padding: index = qx*16 + ((j ^ qx) << 2) + (k ^ qx); the odd element of a
// 32-lane read group flips bit 4.  320 doubles = 2.5 KB per element => 16 waves per CU fit in LDS
// (the padded layout needed 3.2 KB => 12 waves).
template <>
struct NDLayout<3, 4> {
  static constexpr int NC = 4, T = 16, EPW = 4;
  static constexpr int A_FIELD = 64, B_FIELD = 64;
  static constexpr int ELEM = 320, ELEM_PAD = 320;
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) {
    return f * 64 + qx * 16 + (((j ^ qx) & 3) << 2) + ((k ^ qx) & 3);
  }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) {
    return 128 + f * 64 + qx * 16 + (((qy ^ qx) & 3) << 2) + ((k ^ qx) & 3);
  }
  __device__ static __forceinline__ int parity_xor(int sub) { return (sub & 1) << 4; }
};

// The same strides with the two buffers on top of each other (A and B both start at 0): pass Y reads A and writes B, pass
// Y^T reads B and writes A, every lane loads all its inputs before its first store, and the LDS executes a wave's operations
// in order -- so with a compiler fence between the loads and the stores of those two passes (INPLACE below) a lane can never
// see another lane's output where it expects an input.  3/5 of the LDS of the separate buffers (five points per direction,
// where LDS limits the resident waves: pa_nd_hex_stream5.hip).
template <int P1, int Q1>
struct NDLayoutInPlace {
  using S = NDStrides<P1, Q1>;
  static constexpr int NC = P1 + 1;
  static constexpr int A_FIELD = S::Sq * Q1, B_FIELD = S::Tq * Q1;
  static constexpr int ELEM = (2 * A_FIELD > 3 * B_FIELD) ? 2 * A_FIELD : 3 * B_FIELD;
  static constexpr bool INPLACE = true;
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) { return f * A_FIELD + qx * S::Sq + j * S::Sj + k; }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) { return f * B_FIELD + qx * S::Tq + qy * S::Ty + k; }
};
// ... and for four points per direction with the XOR swizzle of NDLayout<3, 4> (p = 3)
struct NDLayoutInPlaceSwz3 {
  static constexpr int NC = 4, A_FIELD = 64, B_FIELD = 64, ELEM = 192, ELEM_PAD = 192;
  static constexpr bool INPLACE = true;
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) {
    return f * 64 + qx * 16 + (((j ^ qx) & 3) << 2) + ((k ^ qx) & 3);
  }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) {
    return f * 64 + qx * 16 + (((qy ^ qx) & 3) << 2) + ((k ^ qx) & 3);
  }
  __device__ static __forceinline__ int parity_xor(int sub) { return (sub & 1) << 4; }
};
template <class LT, class = void>
struct NDInPlace {
  static constexpr bool value = false;
};
template <class LT>
struct NDInPlace<LT, decltype((void)LT::INPLACE)> {
  static constexpr bool value = LT::INPLACE;
};

// ---- forward passes for component C -------------------------------------------------------
template <int C, int P1, int Q1, bool USE_U, bool USE_C, class LT = void, class Args>
__device__ __forceinline__ void nd_fwd_comp(const Args &a, const int e, const bool active,
                                            const bool lane_ok, const int ta, const int tb, const int lx,
                                            double *__restrict__ sm, const double (&u)[P1 + 1],
                                            double (&U)[3][Q1], double (&CU)[3][Q1]) {
  using L = typename std::conditional<std::is_void<LT>::value, NDLayout<P1, Q1>, LT>::type;
  constexpr int NC = L::NC;
  constexpr int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
  const double *TX = (C == 0) ? a.tab.Bo : a.tab.Bc;
  const double *TY = (C == 1) ? a.tab.Bo : a.tab.Bc;
  const double *TZ = (C == 2) ? a.tab.Bo : a.tab.Bc;
  const double *Gc = a.tab.Gc;
  constexpr bool DX = USE_C && C != 0, DY = USE_C && C != 1, DZ = USE_C && C != 2;

  // pass X: lane (j, k) = (ta, tb)
  {
    const bool act = ta < nj && tb < nk;
#pragma unroll
    for (int qx = 0; qx < Q1; qx++) {
      double v = 0.0, d = 0.0;
#pragma unroll
      for (int i = 0; i < ni; i++) {
        v += tab_even<ni, Q1>(TX, qx, i) * u[i];
        if (DX) d += tab_odd<NC, Q1>(Gc, qx, i) * u[i];
      }
      if (lane_ok && act) {
        sm[L::ia(0, qx, ta, tb) ^ lx] = v;
        if (DX) sm[L::ia(1, qx, ta, tb) ^ lx] = d;
      }
    }
  }
  wave_sync();
  // pass Y: lane (qx, k) = (ta, tb)
  {
    const bool act = tb < nk;
    double v[nj], d[nj];
#pragma unroll
    for (int j = 0; j < nj; j++) {
      v[j] = sm[L::ia(0, ta, j, act ? tb : 0) ^ lx];
      if (DX) d[j] = sm[L::ia(1, ta, j, act ? tb : 0) ^ lx];
    }
    if (NDInPlace<L>::value) wave_sync();  // in-place layouts: every lane's loads before any lane's stores
#pragma unroll
    for (int qy = 0; qy < Q1; qy++) {
      double vv = 0.0, vd = 0.0, dv = 0.0;
#pragma unroll
      for (int j = 0; j < nj; j++) {
        vv += tab_even<nj, Q1>(TY, qy, j) * v[j];
        if (DY) vd += tab_odd<NC, Q1>(Gc, qy, j) * v[j];
        if (DX) dv += tab_even<nj, Q1>(TY, qy, j) * d[j];
      }
      if (lane_ok && act) {
        sm[L::ib(0, ta, qy, tb) ^ lx] = vv;
        if (DY) sm[L::ib(1, ta, qy, tb) ^ lx] = vd;
        if (DX) sm[L::ib(2, ta, qy, tb) ^ lx] = dv;
      }
    }
  }
  wave_sync();
  // pass Z: lane (qx, qy) = (ta, tb)
  {
    double vv[nk], vd[nk], dv[nk];
#pragma unroll
    for (int k = 0; k < nk; k++) {
      vv[k] = sm[L::ib(0, ta, tb, k) ^ lx];
      if (DY) vd[k] = sm[L::ib(1, ta, tb, k) ^ lx];
      if (DX) dv[k] = sm[L::ib(2, ta, tb, k) ^ lx];
    }
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      double val = 0.0, dz = 0.0, dy = 0.0, dx = 0.0;
#pragma unroll
      for (int k = 0; k < nk; k++) {
        if (USE_U) val += tab_even<nk, Q1>(TZ, qz, k) * vv[k];
        if (DZ) dz += tab_odd<NC, Q1>(Gc, qz, k) * vv[k];
        if (DY) dy += tab_even<nk, Q1>(TZ, qz, k) * vd[k];
        if (DX) dx += tab_even<nk, Q1>(TZ, qz, k) * dv[k];
      }
      if (USE_U) U[C][qz] = val;
      if (USE_C) {
        // curl(f e_x) = (0, dz f, -dy f); curl(f e_y) = (-dz f, 0, dx f); curl(f e_z) = (dy f, -dx f, 0)
        if (C == 0) CU[1][qz] += dz, CU[2][qz] -= dy;
        if (C == 1) CU[0][qz] -= dz, CU[2][qz] += dx;
        if (C == 2) CU[0][qz] += dy, CU[1][qz] -= dx;
      }
    }
  }
  wave_sync();
}

// ---- transposed passes for component C ------------------------------------------------------
template <int C, int P1, int Q1, bool USE_U, bool USE_C, class LT = void, class Args>
__device__ __forceinline__ void nd_bwd_comp(const Args &a, const int e, const bool active,
                                            const bool lane_ok, const int ta, const int tb, const int lx,
                                            double *__restrict__ sm, double (&rout)[P1 + 1],
                                            const double (&V)[3][Q1], const double (&CV)[3][Q1]) {
  using L = typename std::conditional<std::is_void<LT>::value, NDLayout<P1, Q1>, LT>::type;
  constexpr int NC = L::NC;
  constexpr int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
  const double *TX = (C == 0) ? a.tab.Bo : a.tab.Bc;
  const double *TY = (C == 1) ? a.tab.Bo : a.tab.Bc;
  const double *TZ = (C == 2) ? a.tab.Bo : a.tab.Bc;
  const double *Gc = a.tab.Gc;
  constexpr bool DX = USE_C && C != 0, DY = USE_C && C != 1, DZ = USE_C && C != 2;

  // pass Z^T: lane (qx, qy)
  {
#pragma unroll
    for (int k = 0; k < nk; k++) {
      double vv = 0.0, vd = 0.0, dv = 0.0;
#pragma unroll
      for (int qz = 0; qz < Q1; qz++) {
        // test function curl: C=0 -> (0, dz, -dy); C=1 -> (-dz, 0, dx); C=2 -> (dy, -dx, 0)
        double wdz = 0.0, wdy = 0.0, wdx = 0.0;
        if (USE_C) {
          if (C == 0) wdz = CV[1][qz], wdy = -CV[2][qz];
          if (C == 1) wdz = -CV[0][qz], wdx = CV[2][qz];
          if (C == 2) wdy = CV[0][qz], wdx = -CV[1][qz];
        }
        if (USE_U) vv += tab_even<nk, Q1>(TZ, qz, k) * V[C][qz];
        if (DZ) vv += tab_odd<NC, Q1>(Gc, qz, k) * wdz;
        if (DY) vd += tab_even<nk, Q1>(TZ, qz, k) * wdy;
        if (DX) dv += tab_even<nk, Q1>(TZ, qz, k) * wdx;
      }
      if (lane_ok) {
        sm[L::ib(0, ta, tb, k) ^ lx] = vv;
        if (DY) sm[L::ib(1, ta, tb, k) ^ lx] = vd;
        if (DX) sm[L::ib(2, ta, tb, k) ^ lx] = dv;
      }
    }
  }
  wave_sync();
  // pass Y^T: lane (qx, k)
  {
    const bool act = tb < nk;
    double vv[Q1], vd[Q1], dv[Q1];
#pragma unroll
    for (int qy = 0; qy < Q1; qy++) {
      vv[qy] = sm[L::ib(0, ta, qy, act ? tb : 0) ^ lx];
      if (DY) vd[qy] = sm[L::ib(1, ta, qy, act ? tb : 0) ^ lx];
      if (DX) dv[qy] = sm[L::ib(2, ta, qy, act ? tb : 0) ^ lx];
    }
    if (NDInPlace<L>::value) wave_sync();  // in-place layouts: every lane's loads before any lane's stores
#pragma unroll
    for (int j = 0; j < nj; j++) {
      double v = 0.0, d = 0.0;
#pragma unroll
      for (int qy = 0; qy < Q1; qy++) {
        v += tab_even<nj, Q1>(TY, qy, j) * vv[qy];
        if (DY) v += tab_odd<NC, Q1>(Gc, qy, j) * vd[qy];
        if (DX) d += tab_even<nj, Q1>(TY, qy, j) * dv[qy];
      }
      if (lane_ok && act) {
        sm[L::ia(0, ta, j, tb) ^ lx] = v;
        if (DX) sm[L::ia(1, ta, j, tb) ^ lx] = d;
      }
    }
  }
  wave_sync();
  // pass X^T: lane (j, k), then the signed scatter-add E^T
  {
    const bool act = ta < nj && tb < nk;
    double v[Q1], d[Q1];
#pragma unroll
    for (int qx = 0; qx < Q1; qx++) {
      v[qx] = sm[L::ia(0, qx, act ? ta : 0, act ? tb : 0) ^ lx];
      if (DX) d[qx] = sm[L::ia(1, qx, act ? ta : 0, act ? tb : 0) ^ lx];
    }
#pragma unroll
    for (int i = 0; i < ni; i++) {
      double r = 0.0;
#pragma unroll
      for (int qx = 0; qx < Q1; qx++) {
        r += tab_even<ni, Q1>(TX, qx, i) * v[qx];
        if (DX) r += tab_odd<NC, Q1>(Gc, qx, i) * d[qx];
      }
      rout[i] = r;
    }
  }
  wave_sync();
}

constexpr int kWavesPerBlock = 2;  // small workgroups pack the 160 KB LDS tighter (no workgroup barriers are used)

}  // namespace pa
