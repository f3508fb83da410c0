// Device helpers shared by the element kernels: the reference's pointwise 3x3 utilities and
// coefficient unpacking (fem/qfunctions/33/utils_33_qf.h, fem/qfunctions/coeff/coeff_3_qf.h) and the
// intra-wave LDS hand-off.
#pragma once

#include "pa_internal.hpp"

namespace pa {

__device__ __forceinline__ void wave_sync() {
  // Intra-wave LDS hand-off: the LDS executes a wave's DS operations in order; the fences stop
  // the compiler from moving accesses across, the barrier is a scheduling no-op for one wave.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fem/qfunctions/33/utils_33_qf.h:20-37 (adjugate-transpose, no determinant)
__device__ __forceinline__ void adjJt33(const double J[9], double A[9]) {
  A[0] = J[4] * J[8] - J[7] * J[5];
  A[3] = J[7] * J[2] - J[1] * J[8];
  A[6] = J[1] * J[5] - J[4] * J[2];
  A[1] = J[6] * J[5] - J[3] * J[8];
  A[4] = J[0] * J[8] - J[6] * J[2];
  A[7] = J[3] * J[2] - J[0] * J[5];
  A[2] = J[3] * J[7] - J[6] * J[4];
  A[5] = J[6] * J[1] - J[0] * J[7];
  A[8] = J[0] * J[4] - J[3] * J[1];
}

// utils_33_qf.h:64-84: y = s * A^T B C x (column-major 3x3)
__device__ __forceinline__ void mult_AtBCx33(const double A[9], const double B[9],
                                             const double C[9], const double x0, const double x1,
                                             const double x2, const double s, double &y0,
                                             double &y1, double &y2) {
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

// The same product when the coefficient is c * I (every material isotropic): y = (s c) A^T A x
__device__ __forceinline__ void mult_AtAx33(const double A[9], const double x0, const double x1,
                                            const double x2, const double sc, double &y0, double &y1,
                                            double &y2) {
  const double t0 = A[0] * x0 + A[3] * x1 + A[6] * x2;
  const double t1 = A[1] * x0 + A[4] * x1 + A[7] * x2;
  const double t2 = A[2] * x0 + A[5] * x1 + A[8] * x2;
  y0 = sc * (A[0] * t0 + A[1] * t1 + A[2] * t2);
  y1 = sc * (A[3] * t0 + A[4] * t1 + A[5] * t2);
  y2 = sc * (A[6] * t0 + A[7] * t1 + A[8] * t2);
}

// y = M x for a packed symmetric 3x3 (m = {00, 01, 02, 11, 12, 22})
__device__ __forceinline__ void sym_mv(const double m[6], const double x0, const double x1, const double x2,
                                       double &y0, double &y1, double &y2) {
  y0 = m[0] * x0 + m[1] * x1 + m[2] * x2;
  y1 = m[1] * x0 + m[3] * x1 + m[4] * x2;
  y2 = m[2] * x0 + m[4] * x1 + m[5] * x2;
}

// coeff_3_qf.h:9-24
__device__ __forceinline__ int coeff_index(const CoeffDev &c, int attr) {
  return (c.nattr > 0) ? c.attr_mat[attr - 1] : 0;
}
__device__ __forceinline__ void coeff_unpack3(const CoeffDev &c, int attr, double C[9]) {
  const int k = coeff_index(c, attr);
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = c.mat[9 * k + i];
}


}  // namespace pa
