// Streaming form of the fused E -> B/G -> D -> B^T/G^T -> E^T kernel for Nedelec hexahedra with FIVE points per
// direction (p = 4, BASELINE config 5's element, and its p-coarsened levels) on packed q-data.
//
// Same arithmetic as nd_hex_apply_kernel (pa_nd_hex.hip; reference fem/libceed/operator.cpp:148-178 and the QFunctions
// fem/qfunctions/33/{hdiv_33,hcurl_33,hdivmass_33}_qf.h) and the same schedule as nd_hex_stream_kernel
// (pa_nd_hex_stream.hip): persistent waves walk the batches of one XCD's contiguous element range with a fixed stride and
// keep the HBM streams of the NEXT batch (index block, slot words, x) in flight while they compute the current one; the
// element-local results are signed and stored once, exclusive dofs straight to y, the others to the E-vector that
// et_run_gather_kernel sums by runs.  What differs from the four-point kernel:
//   * one element per 32 lanes (25 columns of five points + 7 lanes that only take part in E / E^T), two elements per wave;
//   * entries m = t + 32 r of the element's sorted dof list per lane, 16-bit slot half-words that carry the orientation /
//     only-copy / essential flags (P = 300 does not fit 8-bit slots), pa_stream_host.hpp: pack_index_wide;
//   * the run-compressed index block (48 words) is fetched with two loads per lane and decoded from LDS;
//   * q-data [ncomp][126] per element: a lane reads its column as two 16-byte pairs and one double (nd_qd_offset).
#include <algorithm>

#include "pa_nd_hex_core.hpp"

namespace pa {

typedef double d2v5 __attribute__((ext_vector_type(2)));

template <int P1>
struct NDStream5Args {
  int ne, nbatch, chunk;  // chunk: batches per XCD (contiguous range)
  const int32_t *blist;   // optional list of batches (interior / interface phases of a multi-rank apply)
  const uint32_t *idxw;   // [nep][kWideWords]
  const uint32_t *perm;   // [nep][NPK][32] slot half-words
  const double *qdata;    // [nep][NG][126]
  const double *coef;     // metric form: [nep][2] scalar mass / curl-curl coefficient of the element
  const double *coef1;    // complex form: the same of the imaginary-part operator
  const double *x, *x1;   // (x1, y1, ye1: the imaginary part of the complex form)
  double *y, *ye, *y1, *ye1;
  // split vectors (SPLIT; NDStreamArgs in pa_nd_hex_stream.hip): ghosts [nsplit, lsize) read from xg0 | xg1 (parity of *xg_sel),
  // written to yg; the three pointers are stored shifted by -nsplit
  int nsplit;
  const double *xg0, *xg1;
  const unsigned long long *xg_sel;
  double *yg;
  NDTab<P1, 5> tab;
};

// QPOS: where the q-data of the batch is requested: 0 at the top of the batch, 1 / 2 after the first / second forward
// component (later = shorter live range of its 60 - 70 registers; curl-curl + mass at p = 4 does not fit 256 otherwise)
// CPLX: y = (A_r + i A_i)(x_r + i x_i) in one pass (pa_op_mult_complex; the complex form of the four-point kernel, DESIGN.md
// 3.1c, carried over): ONE element per wave, its two 32-lane halves hold the real and the imaginary part of x / y; both read
// the same index block and q-data, the parts meet at the D stage (the coefficients are per-element scalars in the metric
// form: the half's own values times the real coefficient -/+ the other half's times the imaginary one).
template <int P1, bool USE_U, bool USE_C, bool METRIC, int GPOS, int QPOS, bool CPLX = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, 2) void nd_hex_stream5_kernel(const NDStream5Args<P1> a) {
  static_assert(!CPLX || (USE_U && USE_C && METRIC), "the complex form is the metric curl-curl + mass kernel");
  static_assert(!(CPLX && SPLIT), "no split-vector form of the complex kernel");
  constexpr int Q1 = 5;
  using L = NDLayoutInPlace<P1, Q1>;  // (LDS limits the resident waves here)
  using streamhost::kWideEss;
  using streamhost::kWideExcl;
  using streamhost::kWideFlip;
  using streamhost::kWideSlotMask;
  using streamhost::kWideStart0;
  using streamhost::kWideWords;
  constexpr int NC = P1 + 1, PP = 3 * P1 * NC * NC, NPL = (PP + 31) / 32, NPK = (NPL + 1) / 2;
  constexpr int NG = METRIC ? (USE_U ? 7 : 6) : 6 * ((USE_U ? 1 : 0) + (USE_C ? 1 : 0));
  constexpr int CS = 126;  // nd_qd_cstride(5)
  constexpr bool EARLY_IDX = P1 < 4;
  // LDS per element (doubles): contraction buffers | element dofs in tensor order (E / E^T staging; its own strip, so that a
  // component's dofs are read just before its forward passes and written back right after its transposed ones instead of
  // all fifteen values per lane living in registers across the batch) | slot words [NPK][32] | index blocks of this and of
  // the next batch [2][48 ints] (the dofs of the exclusive entries are decoded a second time at the E^T stores: cheaper
  // than carrying ten decoded words per lane through the batch)
  constexpr int CONTR_D = (L::ELEM + 1) & ~1, STG_D = (PP + 1) & ~1;
  constexpr int SPW_D = NPK * 16, STAB_D = kWideWords / 2;
  constexpr int LDS_ELEM = CONTR_D + STG_D + SPW_D + 2 * STAB_D;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const int base = xcd * a.chunk, bend = min(base + a.chunk, a.nbatch);
  const int stride = (int)(gridDim.x >> 3) * kWavesPerBlock;
  int k = base + (int)(blockIdx.x >> 3) * kWavesPerBlock + wave;
  if (k >= bend) return;
  int b = a.blist ? a.blist[k] : k;

  // index block and slot words of a batch (arrays padded to whole batches; pad entries read as zero)
  const double *xsel = (CPLX && (lane >> 5)) ? a.x1 : a.x;  // the part of x this half gathers
  const double *xgh = nullptr;  // SPLIT: where the ghost entries are read (shifted: indexed with the local dof)
  if (SPLIT) xgh = ((a.xg_sel ? *a.xg_sel : 0ull) & 1ull) ? a.xg1 : a.xg0;
  auto load_idx = [&](const int bb, const int sub, const int t, unsigned (&w)[2], unsigned (&p)[NPK]) {
    const int e = CPLX ? bb : bb * 2 + sub;
    const uint32_t *ic = a.idxw + (size_t)e * kWideWords;
    w[0] = __builtin_nontemporal_load(&ic[t]);
    w[1] = __builtin_nontemporal_load(&ic[32 + (t & 15)]);
    const uint32_t *pp = a.perm + (size_t)e * (NPK * 32) + t;
#pragma unroll
    for (int q = 0; q < NPK; q++) p[q] = __builtin_nontemporal_load(&pp[32 * q]);
  };
  // dof of entry t + 32 r from an index block in LDS (pa_stream_host.hpp: index_dof_wide)
  auto decode = [&](const int *stab, const int r, const int t) {
    const unsigned bits = (unsigned)stab[2 * r], info = (unsigned)stab[2 * r + 1], low = bits & ((2u << t) - 1u);
    const int rid = (int)(info & 255u) + __popc(low) - 1;
    const int pos = low ? 32 * r + 31 - __clz((int)low) : (int)((info >> 8) & 511u);
    return stab[kWideStart0 + rid] + (t + 32 * r - pos);
  };
  // parks the index block of a batch in LDS, decodes the dofs and requests x of the entries
  auto gather = [&](const unsigned (&w)[2], double (&xv)[NPL], int *stab, const int t) {
    stab[t] = (int)w[0];
    if (t < 16) stab[32 + t] = (int)w[1];
    wave_sync();
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      int dof = decode(stab, r, t);
      if (!(32 * r + 31 < PP) && t + 32 * r >= PP) dof = 0;  // lanes past the last entry
      xv[r] = SPLIT ? (dof < a.nsplit ? xsel : xgh)[dof] : xsel[dof];
    }
  };
  auto settle = [&](unsigned (&p)[NPK]) {
#pragma unroll
    for (int q = 0; q < NPK; q++) asm volatile("" : "+v"(p[q]));
  };
  unsigned wA[2], pA[NPK];
  double xv[NPL];
  int par = 0;  // which of the two index-block strips holds the current batch
  load_idx(b, lane >> 5, lane & 31, wA, pA);
  gather(wA, xv, reinterpret_cast<int *>(smem + (size_t)(wave * 2 + (lane >> 5)) * LDS_ELEM + CONTR_D + STG_D + SPW_D), lane & 31);
#pragma unroll
  for (int r = 0; r < NPL; r++) asm volatile("" : "+v"(xv[r]));
  settle(pA);

  for (;;) {
    // lane constants re-derived from an opaque copy of the lane id in every iteration (hoisted, the LDS addresses and
    // predicates of the passes stay live across the loop and end up in scratch memory)
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int sub = lo >> 5, t = lo & 31;
    const bool lane_ok = t < Q1 * Q1;
    const int tc = lane_ok ? t : 0, ta = tc % Q1, tb = tc / Q1;
    double *sm = smem + (size_t)(wave * 2 + sub) * LDS_ELEM;
    double *stg = sm + CONTR_D;
    int *spw = reinterpret_cast<int *>(stg + STG_D);
    int *stab_cur = spw + 2 * SPW_D + par * kWideWords;
    int *stab = spw + 2 * SPW_D + (par ^ 1) * kWideWords;  // of the next batch
    const int e = CPLX ? b : b * 2 + sub;
    // the next batch (clamped: the last iteration re-reads its own).  Looked up first thing: a batch-list load issued
    // behind the q-data loads would make its use wait for all of them (vmcnt retires in order)
    const int kn = k + stride;
    const bool more = kn < bend;
    const int bn = more ? (a.blist ? a.blist[kn] : kn) : b;

    // q-data of this batch: consumed after the forward contraction (read once: non-temporal)
    d2v5 gq[2 * NG];
    double g4[NG];
    d2v5 ce = {0.0, 0.0}, ci = {0.0, 0.0};
    auto load_qdata = [&]() {
      const double *g = a.qdata + (size_t)e * ((METRIC ? 7 : NG) * CS);  // (the metric form always stores 7 per point)
#pragma unroll
      for (int c = 0; c < NG; c++) {
        const d2v5 *gp = reinterpret_cast<const d2v5 *>(g + c * CS) + tc;
        gq[2 * c] = __builtin_nontemporal_load(&gp[0]);
        gq[2 * c + 1] = __builtin_nontemporal_load(&gp[25]);
        g4[c] = __builtin_nontemporal_load(&g[c * CS + 100 + tc]);
      }
      if (METRIC) ce = reinterpret_cast<const d2v5 *>(a.coef)[e];
      if (CPLX) ci = reinterpret_cast<const d2v5 *>(a.coef1)[e];
    };
    if (QPOS == 0) load_qdata();

    // E: sorted entries into their tensor-order slots (x of this batch was requested during the previous one)
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      if (32 * r + 31 < PP || t + 32 * r < PP) {
        const unsigned h = (pA[r >> 1] >> (16 * (r & 1))) & 0xffffu;
        const double v = (h & kWideEss) ? 0.0 : xv[r];
        stg[h & kWideSlotMask] = (h & kWideFlip) ? -v : v;
      }
    }
#pragma unroll
    for (int q = 0; q < NPK; q++) spw[32 * q + t] = (int)pA[q];
    wave_sync();
    // the lane's line of component C (tensor order) from / to the staging strip
    auto line_in = [&](const int C, double (&u)[NC]) {
      const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
      const bool act = ta < nj && tb < nk;
#pragma unroll
      for (int i = 0; i < NC; i++) u[i] = (act && i < ni) ? stg[C * P1 * NC * NC + i + ni * (ta + nj * tb)] : 0.0;
    };
    auto line_out = [&](const int C, const double (&u)[NC]) {
      const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
      const bool act = lane_ok && ta < nj && tb < nk;
#pragma unroll
      for (int i = 0; i < NC; i++)
        if (act && i < ni) stg[C * P1 * NC * NC + i + ni * (ta + nj * tb)] = u[i];
    };

    // index block of the next batch; first use: the x gather below
    unsigned wB[2], pB[NPK];
    if (EARLY_IDX) {
      load_idx(bn, sub, t, wB, pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    double U[3][Q1], CU[3][Q1];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int q = 0; q < Q1; q++) U[c][q] = 0.0, CU[c][q] = 0.0;
#define PA_S5_FWD(C)                                                                                   \
  {                                                                                                    \
    double ul[NC];                                                                                     \
    line_in(C, ul);                                                                                    \
    nd_fwd_comp<C, P1, Q1, USE_U, USE_C, L>(a, e, true, lane_ok, ta, tb, 0, sm, ul, U, CU);               \
  }
#define PA_S5_BWD(C)                                                                                   \
  {                                                                                                    \
    double ul[NC];                                                                                     \
    nd_bwd_comp<C, P1, Q1, USE_U, USE_C, L>(a, e, true, lane_ok, ta, tb, 0, sm, ul, U, CU);               \
    line_out(C, ul);                                                                                   \
  }
    PA_S5_FWD(0);
    if (QPOS == 1) {
      __builtin_amdgcn_sched_barrier(0);
      load_qdata();
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_S5_FWD(1);
    if (QPOS == 2) {
      __builtin_amdgcn_sched_barrier(0);
      load_qdata();
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_S5_FWD(2);

    if (!EARLY_IDX) {
      __builtin_amdgcn_sched_barrier(0);
      load_idx(bn, sub, t, wB, pB);
      __builtin_amdgcn_sched_barrier(0);
    }

    // D at the five points of this lane's column
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      double H[NG];
#pragma unroll
      for (int c = 0; c < NG; c++) H[c] = qz < 4 ? gq[2 * c + (qz >> 1)][qz & 1] : g4[c];
      if (METRIC) {
        // H = (w / |detJ|) J^T J {00, 01, 02, 11, 12, 22}, H[6] = |detJ| / w:
        //   (w / detJ) J^T c J = c H,   w detJ adj^T c adj = c (|detJ| / w) adj(H)
        double cmass = ce[0], ccurl = ce[1];
        if (CPLX) {
          // (a_r + i a_i)(u_r + i u_i): this half's part of the product, the other part's values from the other half; D is
          // linear in the coefficient, so the geometric matrices below are applied with unit coefficients
          const double sg = sub ? 1.0 : -1.0;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double pu = __shfl_xor(U[c][qz], 32, 64), pcu = __shfl_xor(CU[c][qz], 32, 64);
            U[c][qz] = ce[0] * U[c][qz] + sg * ci[0] * pu;
            CU[c][qz] = ce[1] * CU[c][qz] + sg * ci[1] * pcu;
          }
          cmass = 1.0, ccurl = 1.0;
        }
        if (USE_U) {
          const double cm = H[6] * cmass;
          const double m[6] = {cm * (H[3] * H[5] - H[4] * H[4]), cm * (H[2] * H[4] - H[1] * H[5]), cm * (H[1] * H[4] - H[2] * H[3]),
                               cm * (H[0] * H[5] - H[2] * H[2]), cm * (H[1] * H[2] - H[0] * H[4]), cm * (H[0] * H[3] - H[1] * H[1])};
          sym_mv(m, U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
        }
        if (USE_C) {
          const double m[6] = {ccurl * H[0], ccurl * H[1], ccurl * H[2], ccurl * H[3], ccurl * H[4], ccurl * H[5]};
          sym_mv(m, CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
        }
        __builtin_amdgcn_sched_barrier(0);  // one point at a time: short live ranges
      } else {
        if (USE_U) sym_mv(&H[0], U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
        if (USE_C) sym_mv(&H[USE_U ? 6 : 0], CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
      }
    }

    // x of the next batch: in flight during the (rest of the) transposed passes
    double xB[NPL];
    if (GPOS == 0) {
      __builtin_amdgcn_sched_barrier(0);
      gather(wB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_S5_BWD(0);
    if (GPOS == 1) {
      __builtin_amdgcn_sched_barrier(0);
      gather(wB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_S5_BWD(1);
    if (GPOS == 2) {
      __builtin_amdgcn_sched_barrier(0);
      gather(wB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_S5_BWD(2);

    // E^T: the element-local results (tensor order, in the staging strip) out in sorted order, signed; exclusive dofs
    // straight to y
    wave_sync();
    // one store per entry and lane, unconditionally (the address is selected, not the path: a fixed number of stores keeps
    // the waits for the x values requested before them counted); lanes past the last entry repeat its store
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      const int m = (32 * r + 31 < PP) ? t + 32 * r : min(t + 32 * r, PP - 1), mt = m & 31, mr = m >> 5;
      const unsigned h = ((unsigned)spw[32 * (mr >> 1) + mt] >> (16 * (mr & 1))) & 0xffffu;
      const double v = stg[h & kWideSlotMask];
      int d = decode(stab_cur, mr, mt);
      asm volatile("" : "+v"(d));  // (decoded unconditionally: sunk into a branch otherwise, and the stores stop being counted)
      double *yd = (CPLX && sub) ? a.y1 : a.y;
      if (SPLIT) yd = d < a.nsplit ? a.y : a.yg;
      double *dst = (h & kWideExcl) ? yd + d : ((CPLX && sub) ? a.ye1 : a.ye) + ((size_t)e * PP + m);
      *dst = (h & kWideFlip) ? -v : v;
    }
    wave_sync();  // the LDS strip is reused by the next batch
    if (!more) break;
    k = kn, b = bn, par ^= 1;
#pragma unroll
    for (int r = 0; r < NPL; r++) xv[r] = xB[r];
#pragma unroll
    for (int q = 0; q < NPK; q++) pA[q] = pB[q];
  }
}

#ifndef PA_S5_ONLY  // (register-usage experiments instantiate single kernels without the dispatch below)
// ---- host side ----------------------------------------------------------------------------------------------------------
bool nd_hex_stream5_ok(const SubOp &so) {
  // (read at every operator creation, not cached: tests build both forms in one process)
  if (getenv("PALACE_AMD_STREAM5") && atoi(getenv("PALACE_AMD_STREAM5")) == 0) return false;
  if (so.fe_type != PA_FE_HCURL || so.q1d != 5 || so.p > 4 || !so.qd || !so.d_ye || !so.d_perm_x) return false;
  return so.qd->metric || so.qd->ncomp == 6;
}

static int device_cus5() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

template <int P1, bool U, bool C, bool METRIC, int GPOS, int QPOS, bool CPLX = false, bool SPLIT = false>
static void launch5_gpos(const SubOp &so, NDStream5Args<P1> &a, hipStream_t s) {
  using L = NDLayoutInPlace<P1, 5>;
  for (int i = 0; i < HalfTab<P1, 5>::LEN; i++) a.tab.Bo[i] = so.Bo[i];
  for (int i = 0; i < HalfTab<P1 + 1, 5>::LEN; i++) a.tab.Bc[i] = so.Bc[i], a.tab.Gc[i] = so.Gc[i];
  constexpr int PP = 3 * P1 * (P1 + 1) * (P1 + 1), NPL = (PP + 31) / 32, NPK = (NPL + 1) / 2;
  const size_t lds = sizeof(double) * (size_t)(kWavesPerBlock * 2) * (((L::ELEM + 1) & ~1) + ((PP + 1) & ~1) + NPK * 16 + streamhost::kWideWords);
  const int wg_env = getenv("PALACE_AMD_STREAM_WG") ? atoi(getenv("PALACE_AMD_STREAM_WG")) : 0;
  // workgroups per CU: what the registers (two waves per SIMD) and the LDS admit, and not more than the occupancy query says
  // (with a fixed stride a workgroup that had to queue would run after the others and double the time)
  static const int per_cu_query = [&] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nd_hex_stream5_kernel<P1, U, C, METRIC, GPOS, QPOS, CPLX, SPLIT>, 64 * kWavesPerBlock, lds) !=
            hipSuccess || nb <= 0)
      nb = 4;
    return std::min({nb, (int)(160 * 1024 / lds), 8});
  }();
  const int per_cu = wg_env > 0 ? wg_env : per_cu_query;
  const int per_xcd = std::max(1, device_cus5() / 8) * per_cu;
  if (!a.blist) a.nbatch = CPLX ? so.ne : (so.ne + 1) / 2;  // (complex form: one element per wave)
  if (a.nbatch == 0) return;
  a.chunk = (a.nbatch + 7) / 8;
  int wgx = std::max(1, std::min(per_xcd, (a.chunk + kWavesPerBlock - 1) / kWavesPerBlock));
  // PALACE_AMD_STREAM_WGX caps the workgroups per XCD (tests: many batches per wave on a small mesh)
  if (const char *cap = getenv("PALACE_AMD_STREAM_WGX")) wgx = std::max(1, std::min(wgx, atoi(cap)));
  hipLaunchKernelGGL((nd_hex_stream5_kernel<P1, U, C, METRIC, GPOS, QPOS, CPLX, SPLIT>), dim3(8 * wgx), dim3(64 * kWavesPerBlock), lds, s, a);
  PA_HIP(hipGetLastError());
}

template <int P1, bool U, bool C, bool METRIC>
static void launch5_variant(const SubOp &so, NDStream5Args<P1> &a, hipStream_t s) {
  auto env = [](const char *name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; };
  const int gpos = env("PALACE_AMD_STREAM5_GPOS", U ? 2 : 1);
  if (a.nsplit >= 0) {  // split vectors: the default positions only
    if constexpr (P1 == 4)
      return launch5_gpos<P1, U, C, METRIC, (U ? 2 : 1), ((U && C) ? 2 : 0), false, true>(so, a, s);
    else
      return launch5_gpos<P1, U, C, METRIC, (U ? 2 : 1), 0, false, true>(so, a, s);
  }
  if constexpr (P1 == 4) {  // A/B switches of the order-4 kernels (scripts/time_p4.py)
    const int qpos = env("PALACE_AMD_STREAM5_QPOS", (U && C) ? 2 : 0);
#define PA_S5_LAUNCH(G, Q) launch5_gpos<P1, U, C, METRIC, G, Q>(so, a, s)
#define PA_S5_Q(G) (qpos == 0 ? PA_S5_LAUNCH(G, 0) : qpos == 1 ? PA_S5_LAUNCH(G, 1) : PA_S5_LAUNCH(G, 2))
    gpos == 2 ? PA_S5_Q(2) : PA_S5_Q(1);
  } else {
    if (gpos == 2)
      launch5_gpos<P1, U, C, METRIC, 2, 0>(so, a, s);
    else
      launch5_gpos<P1, U, C, METRIC, 1, 0>(so, a, s);
  }
}

template <int P1>
static void launch5_p(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase, const SplitIO *split,
                      bool all = false) {
  NDStream5Args<P1> a;
  a.nsplit = -1, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr, a.yg = nullptr;
  if (split) {
    PA_REQUIRE(split->n_true >= 0 && split->n_true <= so.lsize, "split point outside the local vector");
    a.nsplit = split->n_true;
    a.xg0 = split->xg0 - split->n_true, a.xg1 = (split->xg1 ? split->xg1 : split->xg0) - split->n_true;
    a.xg_sel = split->sel, a.yg = split->yg - split->n_true;
  }
  a.ne = so.ne;
  a.blist = nullptr, a.nbatch = 0;
  if (phase >= 0) {
    PA_REQUIRE(so.d_blist[phase] || so.n_blist[phase] == 0, "interface batch lists missing");
    a.blist = so.d_blist[phase], a.nbatch = so.n_blist[phase];
    if (a.nbatch == 0) return;
  }
  a.idxw = so.d_idxc;
  a.perm = all ? so.d_perm_s_all : (masked ? so.d_perm_s_bc : so.d_perm_s);  // (all: no entry exclusive -- the fused smoother step)
  a.qdata = so.qd->d;
  a.coef = so.d_coef_s, a.coef1 = nullptr;
  a.x = x, a.y = y, a.ye = so.d_ye;
  a.x1 = nullptr, a.y1 = nullptr, a.ye1 = nullptr;
  const bool m = so.qd->metric;
  switch (so.qf) {
    case PA_QF_HDIV_33:
      if (m) launch5_variant<P1, false, true, true>(so, a, s); else launch5_variant<P1, false, true, false>(so, a, s);
      break;
    case PA_QF_HCURL_33:
      if (m) launch5_variant<P1, true, false, true>(so, a, s); else launch5_variant<P1, true, false, false>(so, a, s);
      break;
    case PA_QF_HDIVMASS_33:
      PA_REQUIRE(m, "streaming curl-curl + mass kernel needs the metric form");
      launch5_variant<P1, true, true, true>(so, a, s);
      break;
    default: throw Error("QFunction not available for H(curl) hexahedra");
  }
}

void launch_nd_hex_stream5(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase, const SplitIO *split,
                           bool all) {
  switch (so.p) {
    case 1: launch5_p<1>(so, x, y, masked, s, phase, split, all); break;
    case 2: launch5_p<2>(so, x, y, masked, s, phase, split, all); break;
    case 3: launch5_p<3>(so, x, y, masked, s, phase, split, all); break;
    case 4: launch5_p<4>(so, x, y, masked, s, phase, split, all); break;
    default: throw Error("no five-point streaming H(curl) hex kernel for this order");
  }
}
template <int P1>
static void launch5_complex_p(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                              double *ye_i, bool masked, hipStream_t s) {
  NDStream5Args<P1> a;
  a.ne = sr.ne, a.blist = nullptr, a.nbatch = 0;
  a.idxw = sr.d_idxc;
  a.perm = masked ? sr.d_perm_s_bc : sr.d_perm_s;
  a.qdata = sr.qd->d;
  a.coef = sr.d_coef_s, a.coef1 = si.d_coef_s;
  a.x = xr, a.x1 = xi, a.y = yr, a.y1 = yi, a.ye = sr.d_ye, a.ye1 = ye_i;
  a.nsplit = -1, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr, a.yg = nullptr;
  if constexpr (P1 == 4)
    launch5_gpos<P1, true, true, true, 2, 2, true>(sr, a, s);
  else
    launch5_gpos<P1, true, true, true, 2, 0, true>(sr, a, s);
}

// the complex form at five points per direction (pa_op_mult_complex; eligibility: nd_hex_stream_complex_ok)
void launch_nd_hex_stream5_complex(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                                   double *ye_i, bool masked, hipStream_t s) {
  switch (sr.p) {
    case 1: launch5_complex_p<1>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    case 2: launch5_complex_p<2>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    case 3: launch5_complex_p<3>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    case 4: launch5_complex_p<4>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    default: throw Error("no five-point streaming H(curl) hex kernel for this order");
  }
}
#endif  // PA_S5_ONLY

}  // namespace pa
