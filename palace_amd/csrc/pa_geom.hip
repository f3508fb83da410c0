// Geometry factors at quadrature points for tensor-product hex meshes.
//
// Replaces the libCEED operator Palace builds in AssembleCeedGeometryData
// (reference fem/libceed/integrator.cpp:335-421) around f_build_geom_factor_33
// (fem/qfunctions/33/geom_33_qf.h:9-33): per point {attr, w*detJ, adj(J)^T/detJ}, stored
// [ne][11][Q] (component-major, point index fastest) exactly like the reference's strided
// CEED_STRIDES_BACKEND q-data (fem/mesh.cpp:188-195).  Set-up only: one thread per point.
#include <algorithm>

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

// Internal element order of a tensor hex block: Morton (Z) order of the element centroids.  The element kernels walk the
// block in this order (eight contiguous ranges, one per XCD, a window of consecutive elements in flight on each), so the
// x / y lines shared by neighbouring elements are re-used from L2 while they are still there; the caller's order (whatever
// the mesh generator produced) only decides where an element's rows sit in the descriptors.  PALACE_AMD_REORDER=0 keeps it.
static std::vector<int32_t> morton_order(const pa_mesh_desc &mesh, int npe) {
  const int ne = mesh.num_elem;
  const char *env = getenv("PALACE_AMD_REORDER");
  if ((env && atoi(env) == 0) || ne < 64) return {};
  std::vector<double> c((size_t)ne * 3, 0.0);
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int e = 0; e < ne; e++) {
    for (int n = 0; n < npe; n++) {
      const int32_t v = mesh.node_offsets[(size_t)e * npe + n];
      for (int d = 0; d < 3; d++) c[(size_t)e * 3 + d] += mesh.nodes[(size_t)v * 3 + d];
    }
    for (int d = 0; d < 3; d++) {
      c[(size_t)e * 3 + d] /= npe;
      lo[d] = std::min(lo[d], c[(size_t)e * 3 + d]), hi[d] = std::max(hi[d], c[(size_t)e * 3 + d]);
    }
  }
  // one cell size for the three directions (the curve should see cubes, not the bounding box's aspect ratio)
  const double span = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300});
  auto spread = [](uint64_t v) {  // 21 bits -> every third bit
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  };
  std::vector<std::pair<uint64_t, int32_t>> key(ne);
  for (int e = 0; e < ne; e++) {
    uint64_t k = 0;
    for (int d = 0; d < 3; d++) {
      const double u = (c[(size_t)e * 3 + d] - lo[d]) / span;
      k |= spread((uint64_t)std::min(2097151.0, std::max(0.0, u * 2097152.0))) << d;
    }
    key[e] = {k, e};
  }
  std::sort(key.begin(), key.end());
  std::vector<int32_t> order(ne);
  for (int e = 0; e < ne; e++) order[e] = key[e].second;
  return order;
}

void launch_geom(const pa_mesh_desc &mesh, Geom &g, hipStream_t s) {
  const int m1 = mesh.mesh_order + 1, npe = m1 * m1 * m1;
  const int ne = mesh.num_elem, q1d = mesh.q1d, Q = q1d * q1d * q1d;
  for (size_t k = 0; k < (size_t)ne * npe; k++)
    PA_REQUIRE(mesh.node_offsets[k] >= 0 && mesh.node_offsets[k] < mesh.num_nodes, "mesh node index out of range");
  g.eorder = morton_order(mesh, npe);
  std::vector<int32_t> off((size_t)ne * npe), attr(ne);
  for (int e = 0; e < ne; e++) {
    const int eo = g.eorder.empty() ? e : g.eorder[e];
    std::copy(mesh.node_offsets + (size_t)eo * npe, mesh.node_offsets + (size_t)(eo + 1) * npe, off.begin() + (size_t)e * npe);
    attr[e] = mesh.attr[eo];
  }
  int32_t *d_off = dev_upload(off.data(), (size_t)ne * npe, s);
  double *d_nodes = dev_upload(mesh.nodes, (size_t)mesh.num_nodes * 3, s);
  int32_t *d_attr = dev_upload(attr.data(), (size_t)ne, s);
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
  if (m1 == 3 && q1d == 4) {  // what the GEOMN form of the streaming kernel reads instead of packed q-data: 648 B per element
    const size_t nep = (size_t)((ne + 3) & ~3);
    std::vector<double> xn(nep * 81, 0.0);
    for (int e = 0; e < ne; e++)
      for (int n = 0; n < 27; n++)
        for (int c = 0; c < 3; c++) xn[((size_t)e * 27 + n) * 3 + c] = mesh.nodes[3 * (size_t)off[(size_t)e * 27 + n] + c];
    for (size_t e = ne; e < nep; e++)  // pad elements: a unit cube (a regular Jacobian: nothing of theirs reaches a result)
      for (int n = 0; n < 27; n++) {
        xn[(e * 27 + n) * 3 + 0] = 0.5 * (n % 3), xn[(e * 27 + n) * 3 + 1] = 0.5 * ((n / 3) % 3), xn[(e * 27 + n) * 3 + 2] = 0.5 * (n / 9);
      }
    g.d_xnodes = dev_upload(xn.data(), xn.size(), s);
    std::vector<double> gt(28);
    for (int q = 0; q < 4; q++) {
      for (int i = 0; i < 3; i++) gt[q * 3 + i] = mesh.mesh_B[q * 3 + i], gt[12 + q * 3 + i] = mesh.mesh_G[q * 3 + i];
      gt[24 + q] = mesh.qweight1d[q];
    }
    g.d_gtab = dev_upload(gt.data(), gt.size(), s);
    PA_HIP(hipStreamSynchronize(s));
  }
  g.d_attr_e = d_attr;
  g.h_attr = attr;
  g.w1.assign(mesh.qweight1d, mesh.qweight1d + q1d);
}

}  // namespace pa
