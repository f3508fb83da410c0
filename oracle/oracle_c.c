/*
 * oracle_c.c — C restatement of the reference's CPU partial-assembly apply.
 *
 * *** TEST INFRASTRUCTURE, not product code: only tests/, __graft_entry__.smoke() and the
 *     cpu_baseline leg of bench.py may load this library. ***
 *
 * What it follows (paths relative to the reference tree):
 *   - the libCEED CPU operator Palace runs on hosts: per element block  E -> dense B -> D -> B^T -> E^T
 *     with the *dense* non-tensor H(curl) tables Palace creates (palace/fem/libceed/basis.cpp:40-85),
 *     oriented restriction (palace/fem/libceed/restriction.cpp:288-298,370-377) and one OpenMP
 *     thread per contiguous element range writing a shared output vector
 *     (palace/fem/mesh.cpp:232-235, palace/fem/libceed/operator.cpp:148-178).  libCEED itself is
 *     not vendored (pin 95bd1e908b16e04a70015e3a9a7fddec5e9c3fc8, cmake/ExternalGitTags.cmake:78-79);
 *     elements are processed in blocks of OC_BLK like its blocked CPU backends.
 *   - geometry factors: palace/fem/qfunctions/33/geom_33_qf.h:9-33
 *   - D: palace/fem/qfunctions/33/{hdiv_33,hcurl_33,hdivmass_33}_qf.h, helpers 33/utils_33_qf.h,
 *     coefficient lookup palace/fem/qfunctions/coeff/{coeff_qf.h,coeff_3_qf.h}
 *
 * Pinned by tests/test_oracle_ref.py (against oracle/_ref = the real headers, and the committed
 * fixtures tests/golden/qf_golden.npz) and, through palace_oracle.py, by the cylinder eig.csv pin.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OC_BLK 8

enum { OC_QF_HDIV = 0, OC_QF_HCURL = 1, OC_QF_HDIVMASS = 2 };

typedef union {
  int32_t first;
  double second;
} oc_int_scalar;

/* coeff_qf.h:18-43 */
static inline int oc_num_attr(const oc_int_scalar *ctx) { return ctx[0].first; }
static inline int oc_num_mat(const oc_int_scalar *ctx) { return ctx[1 + oc_num_attr(ctx)].first; }
static inline const oc_int_scalar *oc_pair_second3(const oc_int_scalar *ctx) {
  return ctx + 2 + oc_num_attr(ctx) + 9 * oc_num_mat(ctx);
}
/* coeff_3_qf.h:9-24 */
static inline void oc_coeff_unpack3(const oc_int_scalar *ctx, int attr, double c[9]) {
  const int k = (oc_num_attr(ctx) > 0) ? ctx[1 + attr - 1].first : 0;
  const oc_int_scalar *m = ctx + 2 + oc_num_attr(ctx);
  for (int i = 0; i < 9; i++) c[i] = m[9 * k + i].second;
}

/* utils_33_qf.h:20-37 */
static inline double oc_adjJt33(const double J[9], double A[9]) {
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

/* utils_33_qf.h:64-84: y = A^T B C x */
static inline void oc_mult_AtBCx33(const double A[9], const double B[9], const double C[9],
                                   const double x[3], double y[3]) {
  double z[3];
  y[0] = C[0] * x[0] + C[3] * x[1] + C[6] * x[2];
  y[1] = C[1] * x[0] + C[4] * x[1] + C[7] * x[2];
  y[2] = C[2] * x[0] + C[5] * x[1] + C[8] * x[2];
  z[0] = B[0] * y[0] + B[3] * y[1] + B[6] * y[2];
  z[1] = B[1] * y[0] + B[4] * y[1] + B[7] * y[2];
  z[2] = B[2] * y[0] + B[5] * y[1] + B[8] * y[2];
  y[0] = A[0] * z[0] + A[1] * z[1] + A[2] * z[2];
  y[1] = A[3] * z[0] + A[4] * z[1] + A[5] * z[2];
  y[2] = A[6] * z[0] + A[7] * z[1] + A[8] * z[2];
}

/* geom_33_qf.h:9-33.  J: [ne][9][Q] (component-major, col-major 3x3), geom: [ne][11][Q]. */
void oc_build_geom_33(int ne, int Q, const double *attr, const double *qw, const double *J,
                      double *geom) {
#pragma omp parallel for schedule(static)
  for (int e = 0; e < ne; e++) {
    const double *Je = J + (size_t)e * 9 * Q;
    double *g = geom + (size_t)e * 11 * Q;
    for (int i = 0; i < Q; i++) {
      double Jl[9], A[9];
      for (int c = 0; c < 9; c++) Jl[c] = Je[c * Q + i];
      const double det = oc_adjJt33(Jl, A);
      g[0 * Q + i] = attr[e];
      g[1 * Q + i] = qw[i] * det;
      for (int c = 0; c < 9; c++) g[(2 + c) * Q + i] = A[c] / det;
    }
  }
}

/* The three QFunctions on one point.  u, cu in; v, cv out. */
static inline void oc_qf_point(int qf, const oc_int_scalar *ctx, const double *g, int Q, int i,
                               const double u[3], const double cu[3], double v[3], double cv[3]) {
  double adj[9], C[9];
  const int attr = (int)g[i];
  const double wdetJ = g[Q + i];
  for (int c = 0; c < 9; c++) adj[c] = g[(2 + c) * Q + i];
  if (qf == OC_QF_HCURL || qf == OC_QF_HDIVMASS) { /* hcurl_33_qf.h:16-27 */
    oc_coeff_unpack3(ctx, attr, C);
    oc_mult_AtBCx33(adj, C, adj, u, v);
    v[0] *= wdetJ, v[1] *= wdetJ, v[2] *= wdetJ;
  }
  if (qf == OC_QF_HDIV || qf == OC_QF_HDIVMASS) { /* hdiv_33_qf.h:16-29, hdivmass_33_qf.h:30-41 */
    double Jl[9];
    oc_coeff_unpack3(qf == OC_QF_HDIVMASS ? oc_pair_second3(ctx) : ctx, attr, C);
    oc_adjJt33(adj, Jl);
    oc_mult_AtBCx33(Jl, C, Jl, cu, cv);
    cv[0] *= wdetJ, cv[1] *= wdetJ, cv[2] *= wdetJ;
  }
}

/* Raw QFunction entry points with the libCEED calling layout ([comp][Q]) — used to compare with
 * oracle/_ref. in_u / in_cu / out_v / out_cv may be NULL when the QFunction does not use them. */
void oc_qfunction(int qf, const void *ctx, int Q, const double *geom, const double *in_u,
                  const double *in_cu, double *out_v, double *out_cv) {
  for (int i = 0; i < Q; i++) {
    double u[3] = {0, 0, 0}, cu[3] = {0, 0, 0}, v[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
    if (in_u) u[0] = in_u[i], u[1] = in_u[Q + i], u[2] = in_u[2 * Q + i];
    if (in_cu) cu[0] = in_cu[i], cu[1] = in_cu[Q + i], cu[2] = in_cu[2 * Q + i];
    oc_qf_point(qf, (const oc_int_scalar *)ctx, geom, Q, i, u, cu, v, cv);
    if (out_v) out_v[i] = v[0], out_v[Q + i] = v[1], out_v[2 * Q + i] = v[2];
    if (out_cv) out_cv[i] = cv[0], out_cv[Q + i] = cv[1], out_cv[2 * Q + i] = cv[2];
  }
}

/*
 * y += E^T B^T D B E x on one element block list.
 *   off [ne][P] int32, ori [ne][P] uint8 (or NULL), interp/deriv [3][Q][P] dense, geom [ne][11][Q].
 * Threads own contiguous element ranges; the scatter into the shared y uses atomics exactly as the
 * reference's threaded CPU path has to (shared output vector, operator.cpp:163-177).
 */
void oc_apply_add(int ne, int P, int Q, const int32_t *off, const uint8_t *ori,
                  const double *interp, const double *deriv, const double *geom, int qf,
                  const void *ctx, const double *x, double *y) {
  const int use_u = (qf == OC_QF_HCURL || qf == OC_QF_HDIVMASS);
  const int use_c = (qf == OC_QF_HDIV || qf == OC_QF_HDIVMASS);
#pragma omp parallel
  {
    double *ue = (double *)malloc(sizeof(double) * P * OC_BLK);
    double *ve = (double *)malloc(sizeof(double) * P * OC_BLK);
    double *uq = (double *)malloc(sizeof(double) * 3 * Q * OC_BLK);
    double *cq = (double *)malloc(sizeof(double) * 3 * Q * OC_BLK);
    double *vq = (double *)malloc(sizeof(double) * 3 * Q * OC_BLK);
    double *wq = (double *)malloc(sizeof(double) * 3 * Q * OC_BLK);
#pragma omp for schedule(static)
    for (int e0 = 0; e0 < ne; e0 += OC_BLK) {
      const int nb = (ne - e0 < OC_BLK) ? ne - e0 : OC_BLK;
      /* E: gather + orientation, interlaced [P][BLK] */
      for (int j = 0; j < P; j++)
        for (int b = 0; b < OC_BLK; b++) {
          double val = 0.0;
          if (b < nb) {
            const size_t k = (size_t)(e0 + b) * P + j;
            val = x[off[k]];
            if (ori && ori[k]) val = -val;
          }
          ue[j * OC_BLK + b] = val;
        }
      /* B: dense [3Q x P] tables */
      for (int r = 0; r < 3 * Q; r++) {
        double su[OC_BLK] = {0}, sc[OC_BLK] = {0};
        if (use_u) {
          const double *t = interp + (size_t)r * P;
          for (int j = 0; j < P; j++)
            for (int b = 0; b < OC_BLK; b++) su[b] += t[j] * ue[j * OC_BLK + b];
        }
        if (use_c) {
          const double *t = deriv + (size_t)r * P;
          for (int j = 0; j < P; j++)
            for (int b = 0; b < OC_BLK; b++) sc[b] += t[j] * ue[j * OC_BLK + b];
        }
        for (int b = 0; b < OC_BLK; b++) uq[r * OC_BLK + b] = su[b], cq[r * OC_BLK + b] = sc[b];
      }
      /* D */
      for (int b = 0; b < nb; b++) {
        const double *g = geom + (size_t)(e0 + b) * 11 * Q;
        for (int i = 0; i < Q; i++) {
          double u[3], cu[3], v[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
          for (int d = 0; d < 3; d++) {
            u[d] = uq[(d * Q + i) * OC_BLK + b];
            cu[d] = cq[(d * Q + i) * OC_BLK + b];
          }
          oc_qf_point(qf, (const oc_int_scalar *)ctx, g, Q, i, u, cu, v, cv);
          for (int d = 0; d < 3; d++) {
            vq[(d * Q + i) * OC_BLK + b] = v[d];
            wq[(d * Q + i) * OC_BLK + b] = cv[d];
          }
        }
      }
      for (int b = nb; b < OC_BLK; b++)
        for (int r = 0; r < 3 * Q; r++) vq[r * OC_BLK + b] = 0.0, wq[r * OC_BLK + b] = 0.0;
      /* B^T */
      memset(ve, 0, sizeof(double) * P * OC_BLK);
      for (int r = 0; r < 3 * Q; r++) {
        if (use_u) {
          const double *t = interp + (size_t)r * P;
          for (int j = 0; j < P; j++)
            for (int b = 0; b < OC_BLK; b++) ve[j * OC_BLK + b] += t[j] * vq[r * OC_BLK + b];
        }
        if (use_c) {
          const double *t = deriv + (size_t)r * P;
          for (int j = 0; j < P; j++)
            for (int b = 0; b < OC_BLK; b++) ve[j * OC_BLK + b] += t[j] * wq[r * OC_BLK + b];
        }
      }
      /* E^T */
      for (int b = 0; b < nb; b++)
        for (int j = 0; j < P; j++) {
          const size_t k = (size_t)(e0 + b) * P + j;
          double val = ve[j * OC_BLK + b];
          if (ori && ori[k]) val = -val;
#pragma omp atomic
          y[off[k]] += val;
        }
    }
    free(ue), free(ve), free(uq), free(cq), free(vq), free(wq);
  }
}

void oc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int oc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
