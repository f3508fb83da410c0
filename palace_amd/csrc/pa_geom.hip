// Geometry factors at quadrature points for tensor-product hex meshes.
//
// Replaces the libCEED operator Palace builds in AssembleCeedGeometryData
// (reference fem/libceed/integrator.cpp:335-421) around f_build_geom_factor_33
// (fem/qfunctions/33/geom_33_qf.h:9-33): per point {attr, w*detJ, adj(J)^T/detJ}, stored
// [ne][11][Q] (component-major, point index fastest) exactly like the reference's strided
// CEED_STRIDES_BACKEND q-data (fem/mesh.cpp:188-195).  Set-up only: one thread per point.
#include "pa_internal.hpp"

namespace pa {

__device__ __forceinline__ double adjJt33_dev(const double J[9], double A[9]) {
  // fem/qfunctions/33/utils_33_qf.h:20-37
  A[0] = J[4] * J[8] - J[7] * J[5];
  A[3] = J[7] * J[2] - J[1] * J[8];
  A[6] = J[1] * J[5] - J[4] * J[2];
  A[1] = J[6] * J[5] - J[3] * J[8];
  A[4] = J[0] * J[8] - J[6] * J[2];
  A[7] = J[3] * J[2] - J[0] * J[5];
  A[2] = J[3] * J[7] - J[6] * J[4];
  A[5] = J[6] * J[1] - J[0] * J[7];
  A[8] = J[0] * J[4] - J[3] * J[1];
  return J[0] * A[0] + J[1] * A[1] + J[2] * A[2];
}

__global__ void geom_factor_kernel(int ne, int q1d, int m1, const int32_t *__restrict__ node_off,
                                   const double *__restrict__ nodes,
                                   const int32_t *__restrict__ attr, const double *__restrict__ B,
                                   const double *__restrict__ G, const double *__restrict__ w1,
                                   double *__restrict__ geom) {
  const int Q = q1d * q1d * q1d;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gid / Q);
  if (e >= ne) return;
  const int q = (int)(gid - (long long)e * Q);
  const int qx = q % q1d, qy = (q / q1d) % q1d, qz = q / (q1d * q1d);
  const int npe = m1 * m1 * m1;
  double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < m1; k++) {
    const double bz = B[qz * m1 + k], gz = G[qz * m1 + k];
    for (int j = 0; j < m1; j++) {
      const double by = B[qy * m1 + j], gy = G[qy * m1 + j];
      for (int i = 0; i < m1; i++) {
        const double bx = B[qx * m1 + i], gx = G[qx * m1 + i];
        const int id = node_off[(size_t)e * npe + i + m1 * (j + m1 * k)];
        const double d0 = gx * by * bz, d1 = bx * gy * bz, d2 = bx * by * gz;
        for (int c = 0; c < 3; c++) {
          const double X = nodes[3 * (size_t)id + c];
          J[c + 0] += X * d0;
          J[c + 3] += X * d1;
          J[c + 6] += X * d2;
        }
      }
    }
  }
  double A[9];
  const double det = adjJt33_dev(J, A);
  double *g = geom + (size_t)e * 11 * Q;
  g[0 * Q + q] = (double)attr[e];
  g[1 * Q + q] = w1[qx] * w1[qy] * w1[qz] * det;
  for (int c = 0; c < 9; c++) g[(2 + c) * Q + q] = A[c] / det;
}

void launch_geom(const pa_mesh_desc &mesh, Geom &g, hipStream_t s) {
  const int m1 = mesh.mesh_order + 1, npe = m1 * m1 * m1;
  const int ne = mesh.num_elem, q1d = mesh.q1d, Q = q1d * q1d * q1d;
  int32_t *d_off = dev_upload(mesh.node_offsets, (size_t)ne * npe, s);
  double *d_nodes = dev_upload(mesh.nodes, (size_t)mesh.num_nodes * 3, s);
  int32_t *d_attr = dev_upload(mesh.attr, (size_t)ne, s);
  double *d_B = dev_upload(mesh.mesh_B, (size_t)q1d * m1, s);
  double *d_G = dev_upload(mesh.mesh_G, (size_t)q1d * m1, s);
  double *d_w = dev_upload(mesh.qweight1d, (size_t)q1d, s);
  g.ne = ne, g.q1d = q1d, g.Q = Q;
  g.d_geom = dev_alloc<double>((size_t)ne * 11 * Q);
  const long long n = (long long)ne * Q;
  const int bs = 256;
  hipLaunchKernelGGL(geom_factor_kernel, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, s, ne,
                     q1d, m1, d_off, d_nodes, d_attr, d_B, d_G, d_w, g.d_geom);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  hipFree(d_off), hipFree(d_nodes), hipFree(d_B), hipFree(d_G), hipFree(d_w);
  g.d_attr_e = d_attr;
  g.h_attr.assign(mesh.attr, mesh.attr + ne);
  g.w1.assign(mesh.qweight1d, mesh.qweight1d + q1d);
}

}  // namespace pa
