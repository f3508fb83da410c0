// Streaming form of the H1 hexahedron apply (y = A x, Q1 = 4: orders 1 .. 3 of the headline hierarchy's auxiliary space):
// the persistent-wave, software-pipelined structure of pa_nd_hex_stream.hip -- fixed stride over per-XCD batch ranges,
// run-compressed element index decoded one batch ahead, x of the next batch requested during the transposed passes,
// one unconditional signed store per entry (exclusive dofs straight to y, the others to the E-vector), E^T of the shared
// dofs by et_run_gather_kernel -- around the sum-factorised H1 contractions of pa_h1_hex.hip (fem/integ/{diffusion,mass,
// diffusionmass}.cpp; D = packed pre-assembled hcurl_33 on grad u / h1_1 / hcurlmass_33).  The one-shot kernel keeps
// y += A x, matrix-free D and the other quadrature sizes.
#include <string>

#include "pa_internal.hpp"

namespace pa {

namespace {

using streamhost::kIdxWords;
constexpr int kIdxStart0 = streamhost::kIdxStart0H1;  // 28 run starts (27 entities of an H1 hexahedron)

constexpr int kH1StreamWaves = 2;

template <int P1>
struct H1StreamTab {  // first two rows of the mirror-symmetric 1-D tables (Q1 = 4)
  double Bc[2 * (P1 + 1)];
  double Gc[2 * (P1 + 1)];
};

template <int N>
__device__ __forceinline__ double hs_even(const double *H, const int q, const int i) {
  return (q < 2) ? H[q * N + i] : H[(3 - q) * N + (N - 1 - i)];
}
template <int N>
__device__ __forceinline__ double hs_odd(const double *H, const int q, const int i) {
  return (q < 2) ? H[q * N + i] : -H[(3 - q) * N + (N - 1 - i)];
}

template <int P1>
struct H1StreamArgs {
  int ne, nbatch, chunk;
  const uint32_t *idxc;  // [ne][kIdxWords]
  const uint32_t *perm;  // [ne][NPK + 1][16]
  const double *qdata;   // [ne][NG][64]
  const double *x;
  double *y, *ye;
  // split vectors (SPLIT, as in pa_nd_hex_stream.hip): local dofs [0, nsplit) live in x / y, the ghosts in xg0 / xg1 (the parity
  // of *xg_sel picks the buffer) and yg, all three stored shifted by -nsplit
  int nsplit;
  const double *xg0, *xg1;
  const unsigned long long *xg_sel;
  double *yg;
  H1StreamTab<P1> tab;
};

__device__ __forceinline__ void hs_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int P1>
struct H1SLayout {  // H1Layout<P1, 4> of pa_h1_hex.hip
  static constexpr int NC = P1 + 1, Q1 = 4;
  static constexpr int A_FIELD = Q1 * NC * NC, B_FIELD = Q1 * Q1 * NC;
  static constexpr int ELEM = 2 * A_FIELD + 3 * B_FIELD;
  static constexpr int ELEM_PAD = ((ELEM + 15) / 16 * 16) | 16;
  __device__ static __forceinline__ int ia(int f, int qx, int j, int k) { return f * A_FIELD + (qx * NC + j) * NC + k; }
  __device__ static __forceinline__ int ib(int f, int qx, int qy, int k) { return 2 * A_FIELD + f * B_FIELD + (qx * Q1 + qy) * NC + k; }
};

template <int P1, bool USE_V, bool USE_G, int MINW, bool SPLIT = false>
__global__ __launch_bounds__(64 * kH1StreamWaves, MINW) void h1_hex_stream_kernel(const H1StreamArgs<P1> a) {
  using L = H1SLayout<P1>;
  constexpr int Q1 = 4, NC = P1 + 1, PP = NC * NC * NC, NPL = (PP + 15) / 16, NPK = (NPL + 3) / 4;
  constexpr int NG = (USE_V ? 1 : 0) + (USE_G ? 6 : 0);
  constexpr int LDS_SIDE = (PP + 1) / 2 + (NPK + 1) * 8;
  // (element stride 16 mod 32 doubles: the two elements of a 32-lane ds_read_b64 group on opposite halves of the banks --
  // see stream_lds_elem in pa_nd_hex_stream.hip)
  constexpr int LDS_ELEM = (L::ELEM_PAD + LDS_SIDE + 14 + 15) / 32 * 32 + 16;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const int base = xcd * a.chunk, bend = min(base + a.chunk, a.nbatch);
  const int stride = (int)(gridDim.x >> 3) * kH1StreamWaves;
  int b = base + (int)(blockIdx.x >> 3) * kH1StreamWaves + wave;
  if (b >= bend) return;
  const double *Bc = a.tab.Bc, *Gc = a.tab.Gc;
  const double *xgh = nullptr;  // SPLIT: where the ghost entries are read
  if (SPLIT) xgh = ((a.xg_sel ? *a.xg_sel : 0ull) & 1ull) ? a.xg1 : a.xg0;

  auto load_idx = [&](const int bb, const int sub, const int t, int (&s)[NPL + 2], unsigned (&p)[NPK + 1]) {
    const int e = bb * 4 + sub;
    const uint32_t *ic = a.idxc + (size_t)e * kIdxWords;
#pragma unroll
    for (int r = 0; r < NPL; r++) s[r] = (int)__builtin_nontemporal_load(&ic[r]);
    s[NPL] = (int)__builtin_nontemporal_load(&ic[kIdxStart0 + t]);
    s[NPL + 1] = (int)__builtin_nontemporal_load(&ic[kIdxStart0 + 16 + min(t, 11)]);
    const uint32_t *pp = a.perm + (size_t)e * ((NPK + 1) * 16) + t;
#pragma unroll
    for (int k = 0; k <= NPK; k++) p[k] = __builtin_nontemporal_load(&pp[16 * k]);
  };
  // decode in place (dof | kEssBit | kExclBit, negative: flipped -- H1 entries never are) and request x
  auto gather = [&](int (&s)[NPL + 2], const unsigned (&p)[NPK + 1], double (&xv)[NPL], int *stab, const int t) {
    stab[t] = s[NPL];
    if (t < 12) stab[16 + t] = s[NPL + 1];
    hs_sync();
    const unsigned fw = p[NPK];
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      const unsigned w = (unsigned)s[r], low = (w & 0xffffu) & ((2u << t) - 1u);
      const int rid = (int)((w >> 16) & 31u) + __popc(low) - 1;
      const int pos = low ? 16 * r + 31 - __clz((int)low) : (int)((w >> 21) & 255u);
      int dof = stab[rid] + (t + 16 * r - pos);
      if (!(16 * r + 15 < PP) && t + 16 * r >= PP) dof = 0;
      xv[r] = SPLIT ? (dof < a.nsplit ? a.x : xgh)[dof] : a.x[dof];
      s[r] = dof | ((fw >> (2 * r + 1)) & 1u ? kExclBit : 0) | ((fw >> (18 + r)) & 1u ? kEssBit : 0);
    }
  };
  auto settle = [&](unsigned (&p)[NPK + 1]) {
#pragma unroll
    for (int k = 0; k <= NPK; k++) asm volatile("" : "+v"(p[k]));
  };
  int sA[NPL + 2];
  unsigned pA[NPK + 1];
  double xv[NPL];
  load_idx(b, lane >> 4, lane & 15, sA, pA);
  gather(sA, pA, xv, reinterpret_cast<int *>(smem + (size_t)(wave * 4 + (lane >> 4)) * LDS_ELEM + L::ELEM_PAD + LDS_SIDE), lane & 15);
#pragma unroll
  for (int r = 0; r < NPL; r++) asm volatile("" : "+v"(xv[r]));
  settle(pA);

  for (;;) {
    int lo = lane;
    asm volatile("" : "+v"(lo));  // lane constants re-derived per batch (see pa_nd_hex_stream.hip)
    const int sub = lo >> 4, t = lo & 15, ta = t & 3, tb = t >> 2;
    double *sm = smem + (size_t)(wave * 4 + sub) * LDS_ELEM;
    int *side = reinterpret_cast<int *>(sm + L::ELEM_PAD);
    int *stab = side + 2 * LDS_SIDE;
    const int e = b * 4 + sub;

    // q-data of this lane's four points (consumed after the forward passes)
    double gd[Q1][NG > 0 ? NG : 1];
    {
      const double *g = a.qdata + (size_t)e * NG * 64 + t;
#pragma unroll
      for (int qz = 0; qz < Q1; qz++)
#pragma unroll
        for (int c = 0; c < NG; c++) gd[qz][c] = __builtin_nontemporal_load(&g[c * 64 + 16 * qz]);
    }
    // index words of the next batch (clamped on the last one)
    const int bn = b + stride;
    const bool more = bn < bend;
    int sB[NPL + 2];
    unsigned pB[NPK + 1];
    load_idx(more ? bn : b, sub, t, sB, pB);
    __builtin_amdgcn_sched_barrier(0);

    // E: sorted entries into their tensor-order slots
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      if (16 * r + 15 < PP || t + 16 * r < PP) {
        const int sv = sA[r];
        sm[(pA[r >> 2] >> (8 * (r & 3))) & 255u] = (sv & kEssBit) ? 0.0 : xv[r];
        side[t + 16 * r] = sv;
      }
    }
#pragma unroll
    for (int k = 0; k <= NPK; k++) side[2 * ((PP + 1) / 2) + 16 * k + t] = (int)pA[k];
    hs_sync();
    double u[NC];
    {
      const bool act = ta < NC && tb < NC;
#pragma unroll
      for (int i = 0; i < NC; i++) u[i] = act ? sm[i + NC * (ta + NC * tb)] : 0.0;
    }
    hs_sync();

    double V[Q1], GV[3][Q1];
    // ---- forward: pass X, lane (j, k)
    {
      const bool act = ta < NC && tb < NC;
#pragma unroll
      for (int qx = 0; qx < Q1; qx++) {
        double v = 0.0, d = 0.0;
#pragma unroll
        for (int i = 0; i < NC; i++) {
          v += hs_even<NC>(Bc, qx, i) * u[i];
          if (USE_G) d += hs_odd<NC>(Gc, qx, i) * u[i];
        }
        if (act) {
          sm[L::ia(0, qx, ta, tb)] = v;
          if (USE_G) sm[L::ia(1, qx, ta, tb)] = d;
        }
      }
    }
    hs_sync();
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
          vv += hs_even<NC>(Bc, qy, j) * v[j];
          if (USE_G) vd += hs_odd<NC>(Gc, qy, j) * v[j];
          if (USE_G) dv += hs_even<NC>(Bc, qy, j) * d[j];
        }
        if (act) {
          sm[L::ib(0, ta, qy, tb)] = vv;
          if (USE_G) sm[L::ib(1, ta, qy, tb)] = vd, sm[L::ib(2, ta, qy, tb)] = dv;
        }
      }
    }
    hs_sync();
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
          if (USE_V) val += hs_even<NC>(Bc, qz, k) * vv[k];
          if (USE_G) {
            dz += hs_odd<NC>(Gc, qz, k) * vv[k];
            dy += hs_even<NC>(Bc, qz, k) * vd[k];
            dx += hs_even<NC>(Bc, qz, k) * dv[k];
          }
        }
        V[qz] = val, GV[0][qz] = dx, GV[1][qz] = dy, GV[2][qz] = dz;
      }
    }
    hs_sync();

    // ---- D: packed pre-assembled h1_1 / hcurl_33 on grad u / hcurlmass_33
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      if (USE_V) V[qz] *= gd[qz][0];
      if (USE_G) {
        const double *m = &gd[qz][USE_V ? 1 : 0];
        const double x0 = GV[0][qz], x1 = GV[1][qz], x2 = GV[2][qz];
        GV[0][qz] = m[0] * x0 + m[1] * x1 + m[2] * x2;
        GV[1][qz] = m[1] * x0 + m[3] * x1 + m[4] * x2;
        GV[2][qz] = m[2] * x0 + m[4] * x1 + m[5] * x2;
      }
    }

    // ---- transposed passes: Z^T lane (qx, qy)
    {
#pragma unroll
      for (int k = 0; k < NC; k++) {
        double vv = 0.0, vd = 0.0, dv = 0.0;
#pragma unroll
        for (int qz = 0; qz < Q1; qz++) {
          if (USE_V) vv += hs_even<NC>(Bc, qz, k) * V[qz];
          if (USE_G) {
            vv += hs_odd<NC>(Gc, qz, k) * GV[2][qz];
            vd += hs_even<NC>(Bc, qz, k) * GV[1][qz];
            dv += hs_even<NC>(Bc, qz, k) * GV[0][qz];
          }
        }
        sm[L::ib(0, ta, tb, k)] = vv;
        if (USE_G) sm[L::ib(1, ta, tb, k)] = vd, sm[L::ib(2, ta, tb, k)] = dv;
      }
    }
    hs_sync();
    // x of the next batch: in flight during the remaining transposed passes
    double xB[NPL];
    __builtin_amdgcn_sched_barrier(0);
    gather(sB, pB, xB, stab, t);
    settle(pB);
    __builtin_amdgcn_sched_barrier(0);
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
          v += hs_even<NC>(Bc, qy, j) * vv[qy];
          if (USE_G) v += hs_odd<NC>(Gc, qy, j) * vd[qy];
          if (USE_G) d += hs_even<NC>(Bc, qy, j) * dv[qy];
        }
        if (act) {
          sm[L::ia(0, ta, j, tb)] = v;
          if (USE_G) sm[L::ia(1, ta, j, tb)] = d;
        }
      }
    }
    hs_sync();
    // X^T lane (j, k)
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
          r += hs_even<NC>(Bc, qx, i) * v[qx];
          if (USE_G) r += hs_odd<NC>(Gc, qx, i) * d[qx];
        }
        u[i] = r;
      }
    }
    hs_sync();
    // E^T: results into tensor order in LDS, out in sorted order; one unconditional store per entry
    {
      const bool act = ta < NC && tb < NC;
#pragma unroll
      for (int i = 0; i < NC; i++)
        if (act) sm[i + NC * (ta + NC * tb)] = u[i];
    }
    hs_sync();
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      const int m = (16 * r + 15 < PP) ? t + 16 * r : min(t + 16 * r, PP - 1), mt = m & 15, mr = m >> 4;
      const unsigned fl = (unsigned)side[2 * ((PP + 1) / 2) + 16 * NPK + mt] >> (2 * mr);
      const double v = sm[((unsigned)side[2 * ((PP + 1) / 2) + 16 * (mr >> 2) + mt] >> (8 * (mr & 3))) & 255u];
      const int d = side[m] & (kExclBit - 1);
      double *yd = a.y;
      if (SPLIT) yd = d < a.nsplit ? a.y : a.yg;
      double *dst = (fl & 2u) ? yd + d : a.ye + ((size_t)e * PP + m);
      *dst = v;
    }
    hs_sync();
    if (!more) break;
    b = bn;
#pragma unroll
    for (int r = 0; r < NPL; r++) sA[r] = sB[r], xv[r] = xB[r];
#pragma unroll
    for (int k = 0; k <= NPK; k++) pA[k] = pB[k];
  }
}

int device_cus_h1() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

template <int P1, bool V, bool G, bool SPLIT = false>
void launch_vg(const SubOp &so, H1StreamArgs<P1> &a, hipStream_t s) {
  using L = H1SLayout<P1>;
  constexpr int MINW = 3;
  constexpr int NC = P1 + 1, PP = NC * NC * NC, NPL = (PP + 15) / 16, NPK = (NPL + 3) / 4;
  for (int i = 0; i < 2 * NC; i++) a.tab.Bc[i] = so.Bc[i], a.tab.Gc[i] = so.Gc[i];
  const size_t lds = sizeof(double) * (size_t)(kH1StreamWaves * 4) * ((L::ELEM_PAD + (PP + 1) / 2 + (NPK + 1) * 8 + 14 + 15) / 32 * 32 + 16);
  static const int per_cu = [&] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, h1_hex_stream_kernel<P1, V, G, MINW, SPLIT>, 64 * kH1StreamWaves, lds) != hipSuccess ||
        nb <= 0)
      nb = 2;
    return std::max(1, std::min(std::min(nb, MINW * 2), 8));
  }();
  a.nbatch = (so.ne + 3) / 4;
  a.chunk = (a.nbatch + 7) / 8;
  const int cus = device_cus_h1();
  // (workgroups are dealt to the XCDs round-robin: the grid is a multiple of 8)
  int grid = std::max(8, std::min(per_cu * cus, ((a.nbatch + kH1StreamWaves - 1) / kH1StreamWaves + 7) / 8 * 8));
  grid = (grid + 7) / 8 * 8;
  hipLaunchKernelGGL((h1_hex_stream_kernel<P1, V, G, MINW, SPLIT>), dim3(grid), dim3(64 * kH1StreamWaves), lds, s, a);
  PA_HIP(hipGetLastError());
}

template <int P1>
void launch_p(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, const SplitIO *split, bool all) {
  H1StreamArgs<P1> a;
  a.ne = so.ne;
  a.idxc = so.d_idxc;
  a.perm = all ? so.d_perm_s_all : (masked ? so.d_perm_s_bc : so.d_perm_s);  // (all: no entry exclusive -- the fused smoother step)
  a.qdata = so.qd->d;
  a.x = x, a.y = y, a.ye = so.d_ye;
  a.nsplit = -1, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr, a.yg = nullptr;
  if (split) {
    PA_REQUIRE(split->n_true >= 0 && split->n_true <= so.lsize, "split point outside the local vector");
    a.nsplit = split->n_true;
    a.xg0 = split->xg0 - split->n_true, a.xg1 = (split->xg1 ? split->xg1 : split->xg0) - split->n_true;
    a.xg_sel = split->sel, a.yg = split->yg - split->n_true;
    switch (so.qf) {
      case PA_QF_HCURL_33: launch_vg<P1, false, true, true>(so, a, s); break;
      case PA_QF_H1_1: launch_vg<P1, true, false, true>(so, a, s); break;
      case PA_QF_HCURLMASS_33: launch_vg<P1, true, true, true>(so, a, s); break;
      default: throw Error("QFunction not available for H1 hexahedra");
    }
    return;
  }
  switch (so.qf) {
    case PA_QF_HCURL_33: launch_vg<P1, false, true>(so, a, s); break;
    case PA_QF_H1_1: launch_vg<P1, true, false>(so, a, s); break;
    case PA_QF_HCURLMASS_33: launch_vg<P1, true, true>(so, a, s); break;
    default: throw Error("QFunction not available for H1 hexahedra");
  }
}

}  // namespace

// the tables of the streaming form can be built (every order the kernel is instantiated for): the split-vector apply of a
// multi-rank ParOperator::Mult always takes this form
bool h1_hex_stream_capable(const SubOp &so) {
  static const bool enabled = [] {
    const char *e = getenv("PALACE_AMD_STREAM"), *h = getenv("PALACE_AMD_STREAM_H1");  // A/B switches
    return !(e && e[0] == '0') && !(h && h[0] == '0');
  }();
  return enabled && so.fe_type == PA_FE_H1 && so.q1d == 4 && so.p >= 1 && so.p <= 3 && so.qd && so.qd->d && so.d_ye && so.d_tptr;
}

// ... and it is the default form of y = A x
bool h1_hex_stream_ok(const SubOp &so) {
  static const bool enabled = [] {
    const char *e = getenv("PALACE_AMD_STREAM"), *h = getenv("PALACE_AMD_STREAM_H1");  // A/B switches
    return !(e && e[0] == '0') && !(h && h[0] == '0');
  }();
  if (!enabled) return false;
  // measured on the 10M-dof hierarchy (125 440 elements): p = 3 diffusion 105 us against 118 us for the one-shot kernel, p = 2 94 us
  // against 78 us (27 entries per element: the fixed per-batch work of the pipeline outweighs what it hides) => p = 3 only by
  // default; PALACE_AMD_STREAM_H1=all takes every order
  static const bool all = [] {
    const char *h = getenv("PALACE_AMD_STREAM_H1");
    return h && std::string(h) == "all";
  }();
  return so.fe_type == PA_FE_H1 && so.q1d == 4 && so.p >= (all ? 1 : 3) && so.p <= 3 && so.qd && so.qd->d && so.d_ye && so.d_tptr;
}

void launch_h1_hex_stream(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, const SplitIO *split, bool all) {
  switch (so.p) {
    case 1: launch_p<1>(so, x, y, masked, s, split, all); break;
    case 2: launch_p<2>(so, x, y, masked, s, split, all); break;
    case 3: launch_p<3>(so, x, y, masked, s, split, all); break;
    default: throw Error("no streaming H1 kernel for this order");
  }
}

}  // namespace pa
