// Streaming form of the fused E -> B/G -> D -> B^T/G^T -> E^T kernel for Nedelec hexahedra with 4 points per
// direction (p = 3 and its p-coarsened levels) on packed q-data, and E^T of the shared dofs by runs.
//
// Same arithmetic and lane mapping as nd_hex_apply_kernel (pa_nd_hex.hip; reference fem/libceed/operator.cpp:148-178
// and the QFunctions fem/qfunctions/33/{hdiv_33,hcurl_33,hdivmass_33}_qf.h), different schedule: the grid is sized to
// what is resident on the chip, every wave walks a sequence of 4-element batches of one XCD's contiguous element range
// and keeps the HBM streams of the NEXT batch in flight while it computes the current one:
//   top of batch i      q-data of batch i (16 B per lane and instruction), index words of batch i + 1
//   forward, D          ...
//   before the transposed passes   x[index] of batch i + 1 (the index words have arrived by now)
//   transposed passes, E^T stores
// so the dependent chain index -> x -> compute that opened every wave of the one-shot kernel (two HBM latencies with
// little else to run) is paid once per wave instead of once per batch.  The index stream is narrower as well: the
// tensor-order slot of a sorted entry is one byte (P <= 256) and the exclusive-dof flag rides in the index word.
//
// E^T: element-local results are signed here and stored to the E-vector in sorted order; dofs with a single copy go
// straight to y.  The shared dofs are then summed by et_run_gather_kernel: because an element's E-vector block is
// sorted by global dof, the copies of consecutive dofs of one mesh entity (a face's 12 dofs, an edge's 3) are
// consecutive in every element that holds them, so the transpose map collapses to runs {first dof, length, first
// position per copy}: 4 bytes of index per dof instead of 12 + 4 per copy, fixed summation order as before.
#include <algorithm>
#include <map>

#include "pa_nd_hex_core.hpp"

#ifndef PA_CPLX_QAHEAD
#define PA_CPLX_QAHEAD 0  // (with PA_STREAM_QAHEAD: also the packed complex form)
#endif
#ifndef PA_STREAM_QAHEAD
#define PA_STREAM_QAHEAD 0  // 1: the instantiations compiled for two waves per SIMD request the q-data one batch ahead
#endif
#ifndef PA_KM_MINW_LOW
#define PA_KM_MINW_LOW 2  // ... and its p < 3 instantiations (the p-coarsened levels); experiment builds: 3
#endif
#ifndef PA_KM_MINW
#define PA_KM_MINW 2  // waves per SIMD the p = 3 curl-curl + mass instantiation is compiled for (experiment builds: 3)
#endif

namespace pa {

typedef double d2v __attribute__((ext_vector_type(2)));

// Timeline instrumentation (timing experiments only, -DPA_STREAM_TRACE): wave 0 of the first kTraceWG workgroups stamps
// s_memtime at the phase boundaries of its first kTraceBatches batches; pa_debug_stream_trace() copies the stamps out.
#ifdef PA_STREAM_TRACE
constexpr int kTraceWG = 64, kTraceBatches = 14, kTraceStamps = 12;
__device__ unsigned long long g_trace[kTraceWG * kTraceBatches * kTraceStamps];
#define PA_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (tr_on && tr_b < kTraceBatches && lane == 0)                                                          \
      g_trace[((size_t)blockIdx.x * kTraceBatches + tr_b) * kTraceStamps + (k)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define PA_STAMP(k) \
  do {              \
  } while (0)
#endif

// Element stride in LDS (doubles).  A wave holds four elements; ds_read_b64 serves lanes 0-31 and 32-63 -- two elements each --
// in one cycle when they touch 32 different 8-byte banks.  Every contraction layout (swizzled p = 3, padded p < 3) spreads
// the 16 lanes of one element over 16 of the 32 banks and its image shifted by 16 doubles over the other 16, so the stride
// has to be 16 (mod 32) doubles.  Rounds 1-3 took the contraction buffers' own stride (ELEM_PAD, which has that property, or
// the parity flip of the swizzled layouts, which assumes 0 mod 32) PLUS the side buffers appended behind them -- 300 doubles
// at p = 3 -- and so ran every second read two- to four-way conflicted: 236 extra LDS cycles per batch on 204
// (scripts/lds_conflict_model.py restates the kernel's access patterns under the guide's banking rules; SQ_LDS_BANK_CONFLICT
// in profiles/r03_apply_pmc.json saw them).
#ifdef PA_STREAM_OLD_LDS  // (A / B builds: the element stride and parity flip of rounds 1-3)
__host__ __device__ constexpr int stream_lds_elem(const int raw) { return raw; }
#else
__host__ __device__ constexpr int stream_lds_elem(const int raw) { return (raw + 15) / 32 * 32 + 16; }
#endif

template <int P1>
struct NDStreamArgs {
  int ne, nbatch, chunk;  // chunk: batches per XCD (contiguous range)
  const int32_t *blist;   // optional list of batches to process (nbatch entries, increasing); NULL: all of 0 .. nbatch - 1
  const uint32_t *idxc;   // [ne][kIdxWords] run-compressed sorted element -> dof index (pa_stream_host.hpp)
  const uint32_t *flagw;  // [ne][16]: bit 2 r = entry t + 16 r is flipped, bit 2 r + 1 = it is the only copy of its dof,
                          // bit 18 + r = essential (read as zero; set in the copy used by masked applies)
  const uint32_t *slots;  // [patterns][NPK][16]: four 8-bit tensor-order slots per word (entries t + 16 r, r = 4 k .. 4 k + 3) of
                          // the sorted -> tensor-order permutation; an element names its pattern in word kIdxPattern of its index
                          // block (elements with the same permutation share one: the table stays in L2)
  const double *qdata;    // [ne][NG][2][16][2]
  const double *qaff;     // [ne][2 NG] pairs: the compact D of affine elements (QData::d_aff); read by the batches whose flag words
                          // carry kAffBit (all four elements affine) instead of qdata
  double wq2[2];          // 1-D quadrature weights {w(0) = w(3), w(1) = w(2)}: the in-plane factor of an affine batch's D
  const double *coef;     // metric form: [ne][2] scalar mass / curl-curl coefficient of the element
  // GEOMN form (geometry from the nodes): [ne][27][3] node coordinates, {B [4][3], G [4][3], w [4]} of the 1-D geometry basis
  const double *xn, *gtab;
  const double *x;
  double *y, *ye;
  // complex form (CPLX): imaginary parts of x, y and of the E-vector, coefficients of the imaginary operator
  const double *x1, *coef1;
  double *y1, *ye1;
  // complex form on packed D (anisotropic materials): q-data of the imaginary operator, and where the mass / curl-curl components of
  // each operator start in its packed block (-1: the operator has no such term), components per point (6 or 12)
  const double *qdata1;
  int qm[2], qc[2], qn[2];
  // split vectors (SPLIT; multi-rank applies without L-vector copies): local dofs [0, nsplit) live in x / y, the ghosts
  // [nsplit, lsize) in xg (input: one of two buffers, chosen by the parity of *xg_sel, a device-resident exchange counter) and
  // yg (output).  xg0 / xg1 / yg are stored shifted by -nsplit, so that they are indexed with the local dof itself.
  int nsplit;
  const double *xg0, *xg1;
  const unsigned long long *xg_sel;
  double *yg;
  NDTab<P1, 4> tab;
};

// GPOS: where x of the next batch is requested (0 before the transposed passes, 1 / 2 / 3 after their first / second /
// third component).  Later = fewer live registers, shorter flight.
// CPLX (metric form only): y = (A_r + i A_i) x for two operators on the same space and geometry whose D differ by the scalar
// coefficients of their elements only -- (a_r + i a_i)(u_r + i u_i) at every quadrature point.  A batch is two elements times
// the two parts of x: the even 16-lane groups of a wave carry the real part, the odd ones the imaginary part of the same
// element, both read the element's index words and q-data (one HBM read), exchange their quadrature values with the
// neighbouring group once and store to the real / imaginary y and E-vector.  One pass over the geometry data instead of four.
// CPLX on packed D (!METRIC; anisotropic materials, round 5): the two operators' symmetric D (mass and / or curl-curl, six doubles per
// point each) do not share a geometric factor, so the even groups load the REAL operator's D and the odd groups the IMAGINARY
// operator's -- every byte of both read once, 96 registers of q-data per lane as in the real kernel -- and each group applies its D
// to BOTH parts of the quadrature values (its own and the neighbouring group's), keeps the product that belongs to its part of y
// and hands the other one over: y_r = D_r u_r - D_i u_i, y_i = D_i u_r + D_r u_i (linalg/operator.cpp:98-134 at the points).
// GEOMN (curl-curl with isotropic coefficients on hex27 elements; round 5): D = (c w / det J) J^T J is recomputed from the 27 nodes
// of the element -- 648 B per element instead of 3 072 B of packed D.  The nodes are requested at the top of the batch (81 doubles
// per element, 6 per lane), parked in the element's LDS strip after the forward passes, and every lane contracts them with its own
// in-plane basis values into 27 partial sums P[k][c][v] (v: d/dxi, d/deta, value; k: node layer), from which the Jacobian at its
// four points along the column costs 27 multiply-adds each.
template <int P1, bool USE_U, bool USE_C, bool METRIC, int MINW, int GPOS, bool CPLX = false, bool SPLIT = false, int GEOMN = 0>
__global__ __launch_bounds__(64 * kWavesPerBlock, MINW) void nd_hex_stream_kernel(const NDStreamArgs<P1> a) {
  static_assert(!CPLX || (USE_U && USE_C), "the complex form is built on the curl-curl + mass kernels");
  static_assert(!(CPLX && SPLIT), "no split-vector form of the complex kernel");
  static_assert(!GEOMN || (!USE_U && USE_C && !METRIC && !CPLX), "geometry from the nodes: the curl-curl kernel");
  constexpr int Q1 = 4;
  // affine batches (round 6): D(q) = w_q D_e read from the compact rows of QData::d_aff (same number of loads, 16 bytes per row
  // for the whole element instead of per lane) and the D stage's result scaled by the in-plane weight; real forms on packed data
  constexpr bool AFF = (!CPLX || METRIC) && !GEOMN && !(PA_STREAM_QAHEAD && MINW == 2);
#ifdef PA_STREAM_EARLY  // experiment builds
  constexpr bool EARLY_IDX = true;
#else
  constexpr bool EARLY_IDX = P1 < 3;
#endif
  // p = 3 curl-curl: the two contraction buffers share their LDS (pa_nd_hex_core.hpp: in-place layout), 2.4 instead of
  // 3.4 KB per element, so that the twelve waves per CU its 168 registers allow are resident instead of ten (measured on the
  // 10M-dof case: 174 -> 167 us per ParOperator::Mult; the kernels with a mass term, two waves per SIMD, gain nothing and the
  // mass kernel loses 2 %, so they keep the separate buffers)
  using L = typename std::conditional<P1 == 3 && !CPLX && USE_C && !USE_U, NDLayoutInPlaceSwz3, NDLayout<P1, Q1>>::type;
  constexpr int NC = P1 + 1, PP = 3 * P1 * NC * NC, NPL = (PP + 15) / 16, NPK = (NPL + 3) / 4;
#ifdef PA_METRIC6  // experiment build: |detJ| / w of the metric form recomputed as w^2 / det(H) instead of read (6 rows instead of 7)
  constexpr int NG = METRIC ? 6 : 6 * ((USE_U ? 1 : 0) + (USE_C ? 1 : 0));
#else
  constexpr int NG = METRIC ? (USE_U ? 7 : 6) : 6 * ((USE_U ? 1 : 0) + (USE_C ? 1 : 0));
#endif
  static_assert(PP <= 256, "8-bit slots");
  using streamhost::kIdxPattern;
  using streamhost::kIdxStart0;
  using streamhost::kIdxWords;
  // LDS per element (doubles): contraction buffers + the batch's index and slot / flag words, kept for the E^T stores,
  // + the run starts of the next batch while its index is decoded
  constexpr int LDS_SIDE = (PP + 1) / 2 + (NPK + 1) * 8;
  constexpr int LDS_XN = GEOMN ? 82 : 0;  // the element's node coordinates (GEOMN)
  constexpr int LDS_ELEM = stream_lds_elem(L::ELEM_PAD + LDS_SIDE + 12 + LDS_XN);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Batches: XCD k (workgroups are dealt to the XCDs round-robin, a placement used for L2 locality only) owns the range
  // [k chunk, (k + 1) chunk); its waves walk it with a fixed stride, so at any time the batches in flight on an XCD are
  // a window of consecutive ones.  (Drawing batches from a counter instead was measured: every memory wait of the
  // kernel tripled with one returning device-scope atomic per wave and batch in the mix, 0.47 vs 0.22 ms.)
  const int xcd = blockIdx.x & 7;
  const int base = xcd * a.chunk, bend = min(base + a.chunk, a.nbatch);
  const int stride = (int)(gridDim.x >> 3) * kWavesPerBlock;
  // (with a batch list -- the interior / interface phases of a multi-rank apply -- positions in the list are walked)
  int k = base + (int)(blockIdx.x >> 3) * kWavesPerBlock + wave;
  if (k >= bend) return;
  int b = a.blist ? a.blist[k] : k;
  // index words of a batch (every array is padded to a multiple of four elements; pad entries read as zero and are
  // stored to E-vector rows nobody gathers)
  // s[0 .. NPL): the slice words (the same word for the 16 lanes of an element), s[NPL], s[NPL + 1]: run starts t, 16 + t
  const double *xsel = (CPLX && ((lane >> 4) & 1)) ? a.x1 : a.x;  // the part of x this 16-lane group gathers
  const double *xgh = nullptr;  // SPLIT: where the ghost entries are read (shifted: indexed with the local dof)
  if (SPLIT) xgh = ((a.xg_sel ? *a.xg_sel : 0ull) & 1ull) ? a.xg1 : a.xg0;
  auto load_idx = [&](const int bb, const int sub, const int t, int (&s)[NPL + 2], unsigned (&p)[NPK + 1]) {
    const int e = CPLX ? bb * 2 + (sub >> 1) : bb * 4 + sub;
    const uint32_t *ic = a.idxc + (size_t)e * kIdxWords;
#pragma unroll
    for (int r = 0; r < NPL; r++) s[r] = (int)__builtin_nontemporal_load(&ic[r]);
    s[NPL] = (int)__builtin_nontemporal_load(&ic[kIdxStart0 + t]);
    s[NPL + 1] = (int)__builtin_nontemporal_load(&ic[kIdxStart0 + 16 + (t & 3)]);
    // flag word now, the slot words when the pattern number has arrived (gather): p[0] carries the number until then
    p[NPK] = __builtin_nontemporal_load(&a.flagw[(size_t)e * 16 + t]);
    p[0] = __builtin_nontemporal_load(&ic[kIdxPattern]);
  };
  // decodes the index of the batch in place (s[r] becomes dof | kEssBit | kExclBit, negative: -(1 + word), flipped) and
  // requests the raw x of the entries (essential entries are zeroed when staged).  stab: 20 ints of LDS of this element.
  auto gather = [&](int (&s)[NPL + 2], unsigned (&p)[NPK + 1], double (&xv)[NPL], int *stab, const int t) {
    stab[t] = s[NPL];
    if (t < 4) stab[16 + t] = s[NPL + 1];
    wave_sync();
    const unsigned fw = p[NPK];
    {  // the slot words of the element's pattern (first use: the E stage of the next batch)
      const uint32_t *sl = a.slots + (size_t)p[0] * (NPK * 16) + t;
#pragma unroll
      for (int k = 0; k < NPK; k++) p[k] = sl[16 * k];
    }
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      const unsigned w = (unsigned)s[r], low = (w & 0xffffu) & ((2u << t) - 1u);
      const int rid = (int)((w >> 16) & 31u) + __popc(low) - 1;
      const int pos = low ? 16 * r + 31 - __clz((int)low) : (int)((w >> 21) & 255u);
      int dof = stab[rid] + (t + 16 * r - pos);
      if (!(16 * r + 15 < PP) && t + 16 * r >= PP) dof = 0;  // lanes past the last entry
      xv[r] = SPLIT ? (dof < a.nsplit ? xsel : xgh)[dof] : xsel[dof];
      const int word = dof | ((fw >> (2 * r + 1)) & 1u ? kExclBit : 0) | ((fw >> (18 + r)) & 1u ? kEssBit : 0);
      s[r] = (fw >> (2 * r)) & 1u ? -1 - word : word;
    }
  };
  // the slot / flag words requested with the index words are first used at the top of the next batch; taking them as
  // arrived here (they were requested before the x values above) keeps that wait from being placed behind the stores
  auto settle = [&](unsigned (&p)[NPK + 1]) { asm volatile("" : "+v"(p[NPK])); };
  int sA[NPL + 2];
  unsigned pA[NPK + 1];
  double xv[NPL];
  load_idx(b, lane >> 4, lane & 15, sA, pA);
  gather(sA, pA, xv,
         reinterpret_cast<int *>(smem + (size_t)(wave * 4 + (lane >> 4)) * LDS_ELEM + L::ELEM_PAD + LDS_SIDE), lane & 15);
  // the first batch's x is awaited here, outside the loop: with loads still pending at the loop entry the compiler merges
  // that state into the loop header and the counted waits at the top of every batch (x requested before that batch's nine
  // stores) degrade to waiting for the stores as well
#pragma unroll
  for (int r = 0; r < NPL; r++) asm volatile("" : "+v"(xv[r]));
  settle(pA);

  // QAHEAD (kernels compiled for two waves per SIMD: they have the registers): the q-data of a batch is requested right after
  // the D stage of the batch before it -- its registers are free from there on -- instead of at the top of its own batch, where
  // only the E stage and the forward passes of the same batch (short at p < 3) stand between the request and the first use
  constexpr bool QAHEAD = PA_STREAM_QAHEAD && MINW == 2 && (!CPLX || (PA_CPLX_QAHEAD && !METRIC));
  d2v gq[GEOMN ? 1 : 2 * NG];
  double xl[GEOMN ? 6 : 1];  // GEOMN: this lane's six of the element's 81 node coordinates, in flight during the forward passes
  auto load_xn = [&](const int ee, const int t) {
    const double *xp = a.xn + (size_t)ee * 81;
#pragma unroll
    for (int r = 0; r < (GEOMN ? 6 : 1); r++) xl[r] = (t + 16 * r < 81) ? __builtin_nontemporal_load(&xp[t + 16 * r]) : 0.0;
  };
  auto load_q = [&](const int ee, const int t, const int aff) {
    if (GEOMN) return load_xn(ee, t);  // (the nodes of the batch: consumed in its D stage)
    if (AFF && !(CPLX && !METRIC)) {
      // one instruction stream for both kinds of batch (the counted waits below depend on the number of loads in flight): the
      // wave-uniform flag selects the base, the lane offset and the row stride
      constexpr int NS = METRIC ? 7 : NG;
      const int rs = aff ? 1 : 16;
      const d2v *g = aff ? reinterpret_cast<const d2v *>(a.qaff) + (size_t)ee * (2 * NS)
                         : reinterpret_cast<const d2v *>(a.qdata) + ((size_t)ee * (2 * NS * 16) + t);
#pragma unroll
      for (int k = 0; k < 2 * NG; k++) gq[k] = __builtin_nontemporal_load(&g[rs * k]);
      return;
    }
    if (CPLX && !METRIC) {  // this group's operator: mass components into gq[0 .. 11], curl-curl into gq[12 .. 23]
      const int grp = (lane >> 4) & 1;
      const int mo = a.qm[grp], co = a.qc[grp];
      const d2v *g = reinterpret_cast<const d2v *>(grp ? a.qdata1 : a.qdata) + ((size_t)ee * (2 * a.qn[grp] * 16) + t);
      const d2v zero = {0.0, 0.0};
#pragma unroll
      for (int k = 0; k < 12; k++) {
        gq[k] = mo >= 0 ? __builtin_nontemporal_load(&g[16 * (2 * mo + k)]) : zero;
        gq[12 + k] = co >= 0 ? __builtin_nontemporal_load(&g[16 * (2 * co + k)]) : zero;
      }
      return;
    }
    const d2v *g = reinterpret_cast<const d2v *>(a.qdata) + ((size_t)ee * (2 * (METRIC ? 7 : NG) * 16) + t);
#pragma unroll
    // (read once: non-temporal, so the stream does not displace x / y lines in L2; measured 5 - 7 % on the apply)
    for (int k = 0; k < (GEOMN ? 1 : 2 * NG); k++) gq[k] = __builtin_nontemporal_load(&g[16 * k]);
  };
  if (QAHEAD) {
    load_q(CPLX ? b * 2 + (lane >> 5) : b * 4 + (lane >> 4), lane & 15, 0);
#pragma unroll
    for (int k = 0; k < 2 * NG; k++) asm volatile("" : "+v"(gq[k]));
  }
#ifdef PA_STREAM_TRACE
  const bool tr_on = blockIdx.x < kTraceWG && wave == 0;
  int tr_b = 0;
#endif
  for (;;) {
    PA_STAMP(0);
    // lane constants are re-derived from an opaque copy of the lane id in every iteration: hoisted out of the loop, the
    // few dozen LDS addresses and predicates of the passes stay live across it and end up in scratch memory
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int sub = lo >> 4, t = lo & 15, ta = t & 3, tb = t >> 2;
    double *sm = smem + (size_t)(wave * 4 + sub) * LDS_ELEM;
    int *side = reinterpret_cast<int *>(sm + L::ELEM_PAD);  // index words of this batch, kept for the E^T stores
    int *stab = side + 2 * LDS_SIDE;                        // run starts of the next batch (decode)
    // (no parity flip of the swizzled layouts here: the element stride itself puts the two elements of a 32-lane read group
    // on opposite halves of the banks, stream_lds_elem)
#ifdef PA_STREAM_OLD_LDS
    const int lx = L::parity_xor(sub);
#else
    constexpr int lx = 0;
#endif
    const int e = CPLX ? b * 2 + (sub >> 1) : b * 4 + sub;

    // q-data of this batch: consumed after the forward contraction
    const int aff = AFF ? __builtin_amdgcn_readfirstlane((int)(pA[NPK] >> 31)) : 0;  // (the same for the 64 lanes: build_stream)
    if (!QAHEAD) load_q(e, t, aff);
    d2v ce = {0.0, 0.0}, ci = {0.0, 0.0};
    if (METRIC || GEOMN) ce = reinterpret_cast<const d2v *>(a.coef)[e];
    if (CPLX && METRIC) ci = reinterpret_cast<const d2v *>(a.coef1)[e];

    // E: sorted entries into their tensor-order slots (x of this batch was requested during the previous one)
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      if (16 * r + 15 < PP || t + 16 * r < PP) {
        const int sv = sA[r], df = sv >= 0 ? sv : -1 - sv;
        const double v = (df & kEssBit) ? 0.0 : xv[r];
        sm[(pA[r >> 2] >> (8 * (r & 3))) & 255u] = sv >= 0 ? v : -v;
        side[t + 16 * r] = sv;
      }
    }
#pragma unroll
    for (int k = 0; k <= NPK; k++) side[2 * ((PP + 1) / 2) + 16 * k + t] = (int)pA[k];
    PA_STAMP(1);  // x arrived, staged
    wave_sync();
    double uin[3][NC];
#pragma unroll
    for (int C = 0; C < 3; C++) {
      const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
      const bool act = ta < nj && tb < nk;
#pragma unroll
      for (int i = 0; i < NC; i++) uin[C][i] = (act && i < ni) ? sm[C * P1 * NC * NC + i + ni * (ta + nj * tb)] : 0.0;
    }
    wave_sync();

    // index words of the next batch (clamped: the last iteration re-reads its own); first use: the x gather below.
    // Requested here when the registers allow (p < 3), after the forward passes otherwise.
    const int kn = k + stride;
    const bool more = kn < bend;
    const int bn = more ? (a.blist ? a.blist[kn] : kn) : b;
    int sB[NPL + 2];
    unsigned pB[NPK + 1];
    if (EARLY_IDX) {
      load_idx(bn, sub, t, sB, pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_STAMP(2);  // q-data requested
    double U[3][Q1], CU[3][Q1];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int q = 0; q < Q1; q++) U[c][q] = 0.0, CU[c][q] = 0.0;
    nd_fwd_comp<0, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[0], U, CU);
    nd_fwd_comp<1, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[1], U, CU);
    nd_fwd_comp<2, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[2], U, CU);

    PA_STAMP(3);  // forward done
    PA_STAMP(4);
    // index words of the next batch, if not requested at the top
    if (!EARLY_IDX) {
      __builtin_amdgcn_sched_barrier(0);
      load_idx(bn, sub, t, sB, pB);
      __builtin_amdgcn_sched_barrier(0);
    }

    // GEOMN: the node coordinates (requested at the top of the batch) into the element's LDS strip, then this lane's partial sums
    // P[k][c][v] over the in-plane nodes of layer k (v: d/dxi, d/deta, value).  GEOMN = 1 keeps the 27 sums in registers (54: the
    // kernel then needs 222-244 and two waves per SIMD).  GEOMN = 2 parks 17 of them in the LDS the element does not use during the
    // D stage -- the contraction buffers between the forward and the transposed passes (12 slots of 16 lanes), the node strip once
    // its last layer has been read (5 slots) -- every lane reading back only what it wrote itself (no synchronisation), and keeps
    // P[.][.][0] and one more in registers.
    constexpr bool PARK = GEOMN == 2;
    double Pz[GEOMN ? 3 : 1][3][PARK ? 1 : 3];
    double Pl = 0.0;  // PARK: P[2][2][2]
    double cw = 0.0;
    double *xs = sm + L::ELEM_PAD + LDS_SIDE + 12;
    // slot q of this lane: the contraction buffers hold 12, the node strip 5 more
    auto park = [&](const int q) -> double * { return (q < 12 ? sm + 16 * q : xs + 16 * (q - 12)) + t; };
    if (GEOMN) {
#pragma unroll
      for (int r = 0; r < 6; r++)
        if (t + 16 * r < 81) xs[t + 16 * r] = xl[r];
      wave_sync();
      const double *gt = a.gtab;
      double bx[3], gx[3], by[3], gy[3];
#pragma unroll
      for (int i = 0; i < 3; i++) bx[i] = gt[ta * 3 + i], gx[i] = gt[12 + ta * 3 + i], by[i] = gt[tb * 3 + i], gy[i] = gt[12 + tb * 3 + i];
      cw = ce[1] * gt[24 + ta] * gt[24 + tb];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        double acc[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c][0] = 0.0, acc[c][1] = 0.0, acc[c][2] = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const double w0 = gx[i] * by[j], w1 = bx[i] * gy[j], w2 = bx[i] * by[j];
            const double *X = xs + 3 * (i + 3 * (j + 3 * k));
#pragma unroll
            for (int c = 0; c < 3; c++) {
              const double xc = X[c];
              acc[c][0] += xc * w0, acc[c][1] += xc * w1, acc[c][2] += xc * w2;
            }
          }
        }
        if (PARK && k == 2) wave_sync();  // (every lane has read the last layer of the nodes: their strip takes parked sums now)
#pragma unroll
        for (int c = 0; c < 3; c++) {
          Pz[k][c][0] = acc[c][0];
          if (!PARK) {
            Pz[k][c][PARK ? 0 : 1] = acc[c][1], Pz[k][c][PARK ? 0 : 2] = acc[c][2];
          } else {
            *park(6 * k + 2 * c) = acc[c][1];
            if (k == 2 && c == 2) Pl = acc[c][2]; else *park(6 * k + 2 * c + 1) = acc[c][2];
          }
        }
        if (PARK) __builtin_amdgcn_sched_barrier(0);  // (a layer at a time: nine accumulators, not twenty-seven)
      }
    }
    // D at the four points of this lane's column
    double wab = 1.0;  // affine batch: the in-plane weight w(ta) w(tb) its compact D leaves out (1: exact no-op otherwise)
    double wxy = 1.0;
    if (AFF) {
      const double wa = (ta == 0 || ta == 3) ? a.wq2[0] : a.wq2[1], wb = (tb == 0 || tb == 3) ? a.wq2[0] : a.wq2[1];
      wxy = wa * wb;
      wab = aff ? wxy : 1.0;
    }
#pragma unroll
    for (int qz = 0; qz < Q1; qz++) {
      double H[NG];
      if (GEOMN) {
        // J[c][d] = d x_c / d xi_d at (ta, tb, qz): the layers combined with the 1-D basis (value for d = 0, 1; derivative for d = 2)
        const double *gt = a.gtab;
        double J[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
          J[c][0] = gt[qz * 3] * Pz[0][c][0] + gt[qz * 3 + 1] * Pz[1][c][0] + gt[qz * 3 + 2] * Pz[2][c][0];
          if (!PARK) {
            J[c][1] = gt[qz * 3] * Pz[0][c][PARK ? 0 : 1] + gt[qz * 3 + 1] * Pz[1][c][PARK ? 0 : 1] + gt[qz * 3 + 2] * Pz[2][c][PARK ? 0 : 1];
            J[c][2] = gt[12 + qz * 3] * Pz[0][c][PARK ? 0 : 2] + gt[12 + qz * 3 + 1] * Pz[1][c][PARK ? 0 : 2] +
                      gt[12 + qz * 3 + 2] * Pz[2][c][PARK ? 0 : 2];
          } else {
            J[c][1] = gt[qz * 3] * *park(2 * c) + gt[qz * 3 + 1] * *park(6 + 2 * c) + gt[qz * 3 + 2] * *park(12 + 2 * c);
            J[c][2] = gt[12 + qz * 3] * *park(2 * c + 1) + gt[12 + qz * 3 + 1] * *park(6 + 2 * c + 1) +
                      gt[12 + qz * 3 + 2] * (c == 2 ? Pl : *park(12 + 2 * c + 1));
          }
        }
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        const double sc = cw * gt[24 + qz] / det;  // (w det J) (J / det)^T c (J / det), hdiv_33_qf.h:10-30 with c I
        H[0] = sc * (J[0][0] * J[0][0] + J[1][0] * J[1][0] + J[2][0] * J[2][0]);
        H[1] = sc * (J[0][0] * J[0][1] + J[1][0] * J[1][1] + J[2][0] * J[2][1]);
        H[2] = sc * (J[0][0] * J[0][2] + J[1][0] * J[1][2] + J[2][0] * J[2][2]);
        H[3] = sc * (J[0][1] * J[0][1] + J[1][1] * J[1][1] + J[2][1] * J[2][1]);
        H[4] = sc * (J[0][1] * J[0][2] + J[1][1] * J[1][2] + J[2][1] * J[2][2]);
        H[5] = sc * (J[0][2] * J[0][2] + J[1][2] * J[1][2] + J[2][2] * J[2][2]);
        if (PARK) {  // one point at a time
          sym_mv(&H[0], CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
      } else {
#pragma unroll
        for (int c = 0; c < NG; c++) H[c] = gq[2 * c + (qz >> 1)][qz & 1];
      }
      if (METRIC) {
        // H = (w / |detJ|) J^T J {00, 01, 02, 11, 12, 22}, H[6] = |detJ| / w:
        //   (w / detJ) J^T c J = c H,   w detJ adj^T c adj = c (|detJ| / w) adj(H)
        double cmass = ce[0], ccurl = ce[1];
        if (CPLX) {
          // (a_r + i a_i)(u_r + i u_i): this group's part of the product, the other part's values from the neighbouring
          // group; D is linear in the coefficient, so the geometric matrices below are applied with unit coefficients
          const double sg = (sub & 1) ? 1.0 : -1.0;
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double pu = __shfl_xor(U[c][qz], 16, 64), pcu = __shfl_xor(CU[c][qz], 16, 64);
            U[c][qz] = ce[0] * U[c][qz] + sg * ci[0] * pu;
            CU[c][qz] = ce[1] * CU[c][qz] + sg * ci[1] * pcu;
          }
          cmass = 1.0, ccurl = 1.0;
        }
        if (USE_U) {
#ifdef PA_METRIC6
          // |detJ| / w = w^2 / det(H)  (det(H) = w^3 / |detJ|); affine rows are H / wab: the factor becomes wz^2 and the common
          // scaling by wab below completes it
          const double c00 = H[3] * H[5] - H[4] * H[4], c01 = H[2] * H[4] - H[1] * H[5], c02 = H[1] * H[4] - H[2] * H[3];
          const double det = H[0] * c00 + H[1] * c01 + H[2] * c02;
          const double wq_ = (aff ? 1.0 : wxy) * ((qz == 0 || qz == 3) ? a.wq2[0] : a.wq2[1]);
          const double cm = cmass * wq_ * wq_ / det;
#else
          const double cm = H[6] * cmass;
#endif
          const double m[6] = {cm * (H[3] * H[5] - H[4] * H[4]), cm * (H[2] * H[4] - H[1] * H[5]), cm * (H[1] * H[4] - H[2] * H[3]),
                               cm * (H[0] * H[5] - H[2] * H[2]), cm * (H[1] * H[2] - H[0] * H[4]), cm * (H[0] * H[3] - H[1] * H[1])};
          sym_mv(m, U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
        }
        if (USE_C) {
          const double m[6] = {ccurl * H[0], ccurl * H[1], ccurl * H[2], ccurl * H[3], ccurl * H[4], ccurl * H[5]};
          sym_mv(m, CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
        }
        if (AFF) {
          // compact rows: H' = H / wab, H'[6] = H[6] wab  =>  both products above came out divided by wab
#pragma unroll
          for (int c = 0; c < 3; c++) {
            if (USE_U) U[c][qz] *= wab;
            if (USE_C) CU[c][qz] *= wab;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one point at a time: short live ranges
      } else if (CPLX) {
        // this group's D on both parts: the product with the real part stays (D_r u_r for y_r, D_i u_r for y_i), the product with
        // the imaginary part goes to the other group (D_i u_i, subtracted from y_r; D_r u_i, added to y_i)
        const bool im = sub & 1;
#pragma unroll
        for (int f = 0; f < 2; f++) {
          double(&W)[3][Q1] = f ? CU : U;
          double ur[3], ui[3], ar[3], ai[3];
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double pv = __shfl_xor(W[c][qz], 16, 64);
            ur[c] = im ? pv : W[c][qz], ui[c] = im ? W[c][qz] : pv;
          }
          sym_mv(&H[6 * f], ur[0], ur[1], ur[2], ar[0], ar[1], ar[2]);
          sym_mv(&H[6 * f], ui[0], ui[1], ui[2], ai[0], ai[1], ai[2]);
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double got = __shfl_xor(ai[c], 16, 64);
            W[c][qz] = im ? ar[c] + got : ar[c] - got;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // one point at a time: short live ranges
      } else {
        if (USE_U) sym_mv(&H[0], U[0][qz], U[1][qz], U[2][qz], U[0][qz], U[1][qz], U[2][qz]);
        if (USE_C) sym_mv(&H[USE_U ? 6 : 0], CU[0][qz], CU[1][qz], CU[2][qz], CU[0][qz], CU[1][qz], CU[2][qz]);
        if (AFF) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            if (USE_U) U[c][qz] *= wab;
            if (USE_C) CU[c][qz] *= wab;
          }
        }
      }
    }

    if (GEOMN == 2) wave_sync();  // (the parked sums have been read: the contraction buffers are the transposed passes' again)
    PA_STAMP(5);  // D done

    if (QAHEAD) {  // q-data of the next batch (the last one re-reads its own)
      __builtin_amdgcn_sched_barrier(0);
      load_q(CPLX ? bn * 2 + (sub >> 1) : bn * 4 + sub, t, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // x of the next batch: in flight during the transposed passes
    double xB[NPL];
    if (GPOS == 0) {
      __builtin_amdgcn_sched_barrier(0);
      gather(sB, pB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    nd_bwd_comp<0, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[0], U, CU);
    PA_STAMP(6);  // first transposed component done
    if (GPOS == 1) {
      __builtin_amdgcn_sched_barrier(0);
      gather(sB, pB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    PA_STAMP(7);  // (GPOS 1) index words landed, x requested
    nd_bwd_comp<1, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[1], U, CU);
    if (GPOS == 2) {
      __builtin_amdgcn_sched_barrier(0);
      gather(sB, pB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    nd_bwd_comp<2, P1, Q1, USE_U, USE_C, L>(a, e, true, true, ta, tb, lx, sm, uin[2], U, CU);
    if (GPOS == 3) {
      __builtin_amdgcn_sched_barrier(0);
      gather(sB, pB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }

    PA_STAMP(8);  // transposed passes done
    // E^T: back to tensor order in LDS, out in sorted order, signed; exclusive dofs straight to y
#pragma unroll
    for (int C = 0; C < 3; C++) {
      const int ni = (C == 0) ? P1 : NC, nj = (C == 1) ? P1 : NC, nk = (C == 2) ? P1 : NC;
      const bool act = ta < nj && tb < nk;
#pragma unroll
      for (int i = 0; i < NC; i++)
        if (act && i < ni) sm[C * P1 * NC * NC + i + ni * (ta + nj * tb)] = uin[C][i];
    }
    wave_sync();
    // One store per entry and lane, unconditionally: exclusive entries (flag) straight to y[dof], the others to the
    // E-vector -- the address is selected, not the path, so the number of stores is fixed and the waits for the x values
    // requested before them can be counted instead of waiting for every store to be acknowledged.  Lanes past the last
    // entry repeat its store.  Essential rows never take the direct path when a fix-up is fused (stream_set_essential
    // routes them through the run gather, which writes x or 0).  (y = A x only: AddMult keeps the one-shot kernel.)
#pragma unroll
    for (int r = 0; r < NPL; r++) {
      const int m = (16 * r + 15 < PP) ? t + 16 * r : min(t + 16 * r, PP - 1), mt = m & 15, mr = m >> 4;
      const unsigned fl = (unsigned)side[2 * ((PP + 1) / 2) + 16 * NPK + mt] >> (2 * mr);
      const double v = sm[((unsigned)side[2 * ((PP + 1) / 2) + 16 * (mr >> 2) + mt] >> (8 * (mr & 3))) & 255u];
      const int sv = side[m], df = sv >= 0 ? sv : -1 - sv, d = df & (kExclBit - 1);
      double *yd = (CPLX && (sub & 1)) ? a.y1 : a.y;
      if (SPLIT) yd = d < a.nsplit ? a.y : a.yg;
      double *dst = (fl & 2u) ? yd + d : ((CPLX && (sub & 1)) ? a.ye1 : a.ye) + ((size_t)e * PP + m);
      *dst = (fl & 1u) ? -v : v;
    }
    wave_sync();  // the LDS strip is reused by the next batch
    PA_STAMP(9);  // stores issued
#ifdef PA_STREAM_TRACE
    tr_b++;
#endif
    if (GPOS == 4) {
      __builtin_amdgcn_sched_barrier(0);
      gather(sB, pB, xB, stab, t);
      settle(pB);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    k = kn, b = bn;
#pragma unroll
    for (int r = 0; r < NPL; r++) sA[r] = sB[r], xv[r] = xB[r];
#pragma unroll
    for (int k = 0; k <= NPK; k++) pA[k] = pB[k];
  }
}

// ---- E^T of the shared dofs by runs -----------------------------------------------------------------------------------
using streamhost::RunHdr;  // {first dof of the run | length - 1 | essential, first entry of its copies in rpos}
using streamhost::RunChunk;

// One thread per shared dof.  Which run a dof belongs to, and where in it, comes from 12 bytes per 64 dofs (RunChunk: a mask of
// the dofs that start a run + the run and offset of the chunk's first dof; popcount and count-leading-zeros as in the element
// index): rounds 1-3 read a code word per dof (run << 4 | offset: 23 MB per apply on the bench mesh).  The essential flag and the
// length ride in the run header.  Every thread walks kGatherILP shared dofs a block width apart with the dependent loads of each
// (header -> copy position -> E-vector) issued side by side: one dof per thread leaves the kernel bound by that chain's latency.
// (Sixteen lanes per run walking the headers alone -- no per-dof data at all -- was measured too: 42.7 against 39.6 us, the idle
// lanes of the short runs cost more instructions than the mask saves.)
constexpr int kGatherILP = 4;
// STEP: the sum is not stored but consumed by a smoother step (GatherStep; y, accumulate, the split arguments unused)
// Two parts in one launch (the one-pass complex apply: the copy positions of the real and the imaginary E-vector are the same):
// ye1 / y1 / x1 != nullptr sums the second E-vector into y1 alongside, every header and position word read once.
struct GatherPart2 {
  const double *ye1, *x1;
  double *y1;
};
template <bool STEP, bool DUAL = false>
__global__ __launch_bounds__(256) void et_run_gather_kernel_t(const int n, const RunChunk *__restrict__ chunk,
                                                              const RunHdr *__restrict__ hdr, const int32_t *__restrict__ rpos,
                                                              const double *__restrict__ ye, double *__restrict__ y,
                                                              const int accumulate, const double *__restrict__ x,
                                                              const int ess_policy, const int nsplit, double *__restrict__ yg,
                                                              const GatherStep st, const GatherPart2 p2) {
  const int k0 = blockIdx.x * (256 * kGatherILP) + threadIdx.x;
  const int lane = threadIdx.x & 63;
  RunHdr h[kGatherILP];
  int pe[kGatherILP], d[kGatherILP], j[kGatherILP], run[kGatherILP];
  bool live[kGatherILP], fix[kGatherILP];
  double s[kGatherILP], yold[kGatherILP];
#pragma unroll
  for (int u = 0; u < kGatherILP; u++) {
    const int k = k0 + 256 * u;
    live[u] = k < n;
    const RunChunk c = chunk[live[u] ? (k >> 6) : 0];  // (the same 16 bytes for the 64 lanes of a wave)
    const unsigned long long low = (c.starts & ~1ull) & ((2ull << lane) - 1ull);  // run starts in (first dof of the chunk, this dof]
    const int nc = __popcll(low);
    run[u] = (int)(c.first >> 4) + nc;
    j[u] = nc ? lane - (63 - __clzll((long long)low)) : (int)(c.first & 15u) + lane;
  }
#pragma unroll
  for (int u = 0; u < kGatherILP; u++) {
    h[u] = hdr[live[u] ? run[u] : 0];
    pe[u] = hdr[live[u] ? run[u] + 1 : 0].ptr;
    fix[u] = (h[u].dof0 >> 31) && ess_policy >= 0;
  }
#pragma unroll
  for (int u = 0; u < kGatherILP; u++) {
    d[u] = (int)(h[u].dof0 & streamhost::kRunDofMask) + j[u];
    s[u] = 0.0;
    yold[u] = 0.0;
    if (live[u] && fix[u]) {
      if (ess_policy) s[u] = x[d[u]];
      pe[u] = h[u].ptr;  // no copies to sum
    } else if (!STEP && live[u] && accumulate) {
      yold[u] = y[d[u]];
    }
  }
  // STEP: the vectors of the recurrence, requested with the copies
  double se[STEP ? kGatherILP : 1], sdi[STEP ? kGatherILP : 1], sr0[STEP ? kGatherILP : 1], sep[STEP ? kGatherILP : 1],
      so[STEP ? kGatherILP : 1];
  if (STEP) {
#pragma unroll
    for (int u = 0; u < kGatherILP; u++) {
      const int dd = (live[u] && d[u] < nsplit) ? d[u] : 0;  // (ghost rows -- split form -- take no part in the recurrence)
      sr0[u] = st.r0[dd];
      sdi[u] = st.dinv ? st.dinv[dd] : 0.0;
      se[u] = st.mode == 1 ? x[dd] : 0.0;
      sep[u] = (st.mode == 1 && st.ep) ? st.ep[dd] : 0.0;
      so[u] = (st.mode == 1 && st.add) ? st.out[dd] : 0.0;
    }
  }
  // copies in order (fixed summation order; an absent copy adds an exact zero): the first four of every dof side by
  // side -- interior faces have 2, edges 4 -- then the rare rest one at a time
  int pos[kGatherILP][4];
#pragma unroll
  for (int u = 0; u < kGatherILP; u++)
#pragma unroll
    for (int q = 0; q < 4; q++) pos[u][q] = (live[u] && h[u].ptr + q < pe[u]) ? rpos[h[u].ptr + q] : -1;
  double v[kGatherILP][4];
#pragma unroll
  for (int u = 0; u < kGatherILP; u++)
#pragma unroll
    for (int q = 0; q < 4; q++) v[u][q] = pos[u][q] >= 0 ? ye[(size_t)pos[u][q] + j[u]] : 0.0;
  if (DUAL) {  // the second part: same positions, its own E-vector
    double w[kGatherILP][4], s1[kGatherILP];
#pragma unroll
    for (int u = 0; u < kGatherILP; u++)
#pragma unroll
      for (int q = 0; q < 4; q++) w[u][q] = pos[u][q] >= 0 ? p2.ye1[(size_t)pos[u][q] + j[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kGatherILP; u++) {
      s1[u] = (live[u] && fix[u] && ess_policy) ? p2.x1[d[u]] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) s1[u] += w[u][q];
      if (live[u])
        for (int p = h[u].ptr + 4; p < pe[u]; p++) s1[u] += p2.ye1[(size_t)rpos[p] + j[u]];
      if (live[u]) p2.y1[d[u]] = s1[u];
    }
  }
#pragma unroll
  for (int u = 0; u < kGatherILP; u++) {
#pragma unroll
    for (int q = 0; q < 4; q++) s[u] += v[u][q];
    if (live[u])
      for (int p = h[u].ptr + 4; p < pe[u]; p++) s[u] += ye[(size_t)rpos[p] + j[u]];
  }
  // (split vectors: rows [nsplit, ...) are ghosts and go to yg, stored shifted by -nsplit; nsplit = INT_MAX otherwise)
  if (STEP) {
    // OpChebStep3 (linalg.hip) with t = s: out (+)= e + sd (e - e_prev) + sr dinv (r0 - t)
#pragma unroll
    for (int u = 0; u < kGatherILP; u++) {
      if (!live[u]) continue;
      if (d[u] >= nsplit) {  // a ghost row: to its owner through the halo kernel
        yg[d[u]] = s[u];
        continue;
      }
      if (st.iface_mask && (st.iface_mask[d[u]] & 2)) {  // an owned dof with sharers: the halo kernel completes the sum and the step
        st.t_iface[d[u]] = s[u];
        continue;
      }
      const double rv = sr0[u] - s[u];
      if (st.mode == 2) {  // residual (and the first direction of the polynomial)
        if (st.res) st.res[d[u]] = rv;
        if (st.out) st.out[d[u]] = st.sr * sdi[u] * rv;
        continue;
      }
      double dk = st.sr * sdi[u] * rv;
      dk += st.sd * (se[u] - sep[u]);
      st.out[d[u]] = so[u] + (se[u] + dk);
    }
    return;
  }
#pragma unroll
  for (int u = 0; u < kGatherILP; u++)
    if (live[u]) (d[u] < nsplit ? y : yg)[d[u]] = yold[u] + s[u];
}

// ---- host side ----------------------------------------------------------------------------------------------------------
bool nd_hex_stream_ok(const SubOp &so) {
  static const bool enabled = !(getenv("PALACE_AMD_STREAM") && atoi(getenv("PALACE_AMD_STREAM")) == 0);
  if (!enabled) return false;
  if (so.q1d == 5) return nd_hex_stream5_ok(so);  // five points per direction: pa_nd_hex_stream5.hip
  if (so.fe_type != PA_FE_HCURL || so.q1d != 4 || so.p > 3 || !so.qd || !so.d_ye || !so.d_perm_x) return false;
  return so.qd->metric || so.qd->ncomp == 6 || (so.qd->ncomp == 12 && so.qf == PA_QF_HDIVMASS_33);
}

static bool wide_form(const SubOp &so) { return so.fe_type == PA_FE_HCURL && so.q1d == 5; }

// Scalar mass / curl-curl coefficient of every element (coeff_3_qf.h:9-24 resolved on the host; isotropic materials): what the
// metric form multiplies its geometric matrices with, [ne padded to 4][2]
void stream_element_coefficients(SubOp &so) {
  if (so.d_coef_s) return;
  const int ne = so.ne, nep = (ne + 3) & ~3;
  const std::vector<int32_t> &attr = so.geom->h_attr;
  PA_REQUIRE((int)attr.size() == ne, "element attributes missing");
  const CoeffHost *cm = nullptr, *cc = nullptr;
  if (so.qf == PA_QF_HDIV_33) cc = &so.c0;
  if (so.qf == PA_QF_HCURL_33) cm = &so.c0;
  if (so.qf == PA_QF_HDIVMASS_33) cm = &so.c0, cc = &so.c1;
  auto value = [&](const CoeffHost *c, int at) {
    if (!c) return 0.0;
    int k = 0;
    if (!c->attr_mat.empty()) {
      PA_REQUIRE(at >= 1 && at <= (int)c->attr_mat.size(), "element attribute outside the coefficient's attribute map");
      k = c->attr_mat[at - 1];
    }
    return c->mat[(size_t)9 * k];
  };
  std::vector<double> coef((size_t)nep * 2, 0.0);
  for (int e = 0; e < ne; e++) coef[2 * (size_t)e] = value(cm, attr[e]), coef[2 * (size_t)e + 1] = value(cc, attr[e]);
  so.d_coef_s = dev_upload(coef.data(), coef.size());
}

// Affine elements: one wave per element looks at the packed q-data [ncomp][2][16][2] (nd_qd_offset, four points per direction).
// Component c is w_q r_c(q) (the metric form's component 6: r / w_q); the element is affine when every r_c is the same at the 64
// points to `tol` relative to the largest |mean r| of its group of six (mass / curl-curl / metric matrix; component 6 on its own).
// Writes the compact rows (means) and the flag.
__global__ __launch_bounds__(64) void stream_affine_kernel(const int ne, const int ncomp, const int metric, const double *__restrict__ qd,
                                                           const double w0, const double w1, const double tol,
                                                           double *__restrict__ aff, unsigned char *__restrict__ flag) {
  const int e = blockIdx.x, q = threadIdx.x;
  if (e >= ne) return;
  auto w1d = [&](const int i) { return (i == 0 || i == 3) ? w0 : w1; };
  const double wq = w1d(q & 3) * w1d((q >> 2) & 3) * w1d(q >> 4);
  auto wave_sum = [](double v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
  };
  auto wave_max = [](double v) {
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
  };
  bool ok = true;
  for (int g0 = 0; g0 < ncomp; g0 += 6) {
    const int gn = min(6, ncomp - g0);
    double mean[6], dev = 0.0, scale = 0.0;
    for (int c = 0; c < gn; c++) {
      const double v = qd[(size_t)e * ncomp * 64 + nd_qd_offset(4, g0 + c, q)];
      const double r = (metric && g0 + c == 6) ? v * wq : v / wq;
      mean[c] = wave_sum(r) * (1.0 / 64.0);
      dev = fmax(dev, wave_max(fabs(r - mean[c])));
      scale = fmax(scale, fabs(mean[c]));
    }
    ok = ok && dev <= tol * scale;
    if (q < 2 * gn) {  // row 2 c + h: {r wz(2h), r wz(2h + 1)}, component 6 of the metric form: r / wz
      const int c = q >> 1, h = q & 1;
      const bool inv = metric && g0 + c == 6;
      double *row = aff + ((size_t)e * 2 * ncomp + 2 * (g0 + c) + h) * 2;
      double m = 0.0;
      for (int k = 0; k < 6; k++) m = (k == c) ? mean[k] : m;  // (no dynamic register indexing)
      row[0] = inv ? m / w1d(2 * h) : m * w1d(2 * h);
      row[1] = inv ? m / w1d(2 * h + 1) : m * w1d(2 * h + 1);
    }
  }
  if (q == 0) flag[e] = ok ? 1 : 0;
}

// Once per QData (the p-coarsened operators share it): which elements are affine, which batches of four consist of such
// elements only, the compact rows.  PALACE_AMD_STREAM_AFFINE=0 switches the form off (A / B runs; read when the data is built).
static void stream_affine_setup(SubOp &so) {
  QData &qd = *so.qd;
  if (qd.aff_done) return;
  qd.aff_done = true;
  const bool enabled = !(getenv("PALACE_AMD_STREAM_AFFINE") && atoi(getenv("PALACE_AMD_STREAM_AFFINE")) == 0);
  if (!enabled || so.q1d != 4 || (int)so.geom->w1.size() != 4) return;
  const std::vector<double> &w = so.geom->w1;
  if (w[0] != w[3] || w[1] != w[2]) return;  // (not a symmetric rule: keep the per-point data)
  const int ne = so.ne, nep = (ne + 3) & ~3;
  const size_t nrow = (size_t)nep * 2 * qd.ncomp * 2;
  double *d_aff = nullptr;
  unsigned char *d_flag = nullptr;
  PA_HIP(hipMalloc(&d_aff, nrow * sizeof(double)));
  PA_HIP(hipMemset(d_aff, 0, nrow * sizeof(double)));
  PA_HIP(hipMalloc(&d_flag, (size_t)nep));
  PA_HIP(hipMemset(d_flag, 0, (size_t)nep));
  const double tol = getenv("PALACE_AMD_AFFINE_TOL") ? atof(getenv("PALACE_AMD_AFFINE_TOL")) : 1e-13;
  hipLaunchKernelGGL(stream_affine_kernel, dim3(ne), dim3(64), 0, nullptr, ne, qd.ncomp, qd.metric ? 1 : 0, qd.d, w[0], w[1], tol, d_aff,
                     d_flag);
  PA_HIP(hipGetLastError());
  std::vector<unsigned char> flag((size_t)nep, 0);
  PA_HIP(hipMemcpy(flag.data(), d_flag, (size_t)nep, hipMemcpyDeviceToHost));
  (void)hipFree(d_flag);
  qd.batch_aff.assign((size_t)nep / 4, 0);
  for (int e = 0; e < ne; e++) qd.n_aff_elems += flag[e];
  for (int b = 0; b < nep / 4; b++)
    if (flag[4 * b] && flag[4 * b + 1] && flag[4 * b + 2] && flag[4 * b + 3]) qd.batch_aff[b] = 1, qd.n_aff_batch_elems += 4;
  if (qd.n_aff_batch_elems == 0) {
    (void)hipFree(d_aff);
    qd.batch_aff.clear();
    return;
  }
  qd.d_aff = d_aff;
  qd.wq2[0] = w[0], qd.wq2[1] = w[1];
}

// Index arrays of the streaming kernel and the run form of the transpose map (after finalize_exclusive: needs the
// exclusive flags).  Host work proportional to the index array, once per operator.  Every per-element array is padded
// to a multiple of four elements (one batch); the pad entries are flagged essential (read as zero).
void build_stream(SubOp &so) {
  if (so.d_idxc || !(nd_hex_stream_ok(so) || h1_hex_stream_capable(so))) return;
  const int P = so.P, ne = so.ne;
  std::vector<uint32_t> ic, pp;
  // (a numbering that breaks an element's dofs into more than kIdxMaxRuns runs keeps the one-shot kernel)
  if (wide_form(so)) {
    if (!streamhost::pack_index_wide(ne, P, so.lsize, so.h_sidx.data(), so.h_perm.data(), ic, pp)) return;
  } else if (!streamhost::pack_index(ne, P, so.lsize, so.h_sidx.data(), so.h_perm.data(), ic, pp,
                                     so.fe_type == PA_FE_H1 ? streamhost::kIdxStart0H1 : streamhost::kIdxStart0))
    return;
  // PRICING EXPERIMENT (wrong results, right bytes; never set in product runs): PALACE_AMD_PRICE_BLOCK=G with G = 4 (the elements of
  // one wave) or 8 (of one workgroup) prices an E-vector in which the copies of a dof inside a group of G consecutive elements
  // are assembled on chip before the store: every copy but the group's first is taken off the E-vector (its store goes
  // straight to y instead, like an exclusive dof's) and out of the run lists; a dof whose copies all sit in one group leaves
  // the gather altogether.  What is NOT priced: the LDS traffic and barrier of the on-chip assembly itself.
#ifdef PA_ABLATION  // (the ablation library only, `make ablate`: the product library has no such switch)
  const int price_group = (so.fe_type == PA_FE_HCURL && !wide_form(so) && getenv("PALACE_AMD_PRICE_BLOCK")) ? atoi(getenv("PALACE_AMD_PRICE_BLOCK")) : 0;
#else
  constexpr int price_group = 0;
#endif
  if (price_group > 1) {
    const size_t nnz = (size_t)ne * P;
    const int npl = (P + 15) / 16, npk = (npl + 3) / 4;
    so.h_price_skip.assign(nnz, 0);
    std::vector<int32_t> seen((size_t)so.lsize, -1), cnt((size_t)so.lsize, 0), cnt2((size_t)so.lsize, 0);
    for (size_t k = 0; k < nnz; k++) {
      const int d = streamhost::dof_of(so.h_sidx[k]), g = (int)(k / P) / price_group;
      cnt[d]++;
      if (seen[d] == g) so.h_price_skip[k] = 1; else seen[d] = g, cnt2[d]++;
    }
    size_t dropped = 0, freed = 0;
    for (size_t k = 0; k < nnz; k++) {
      const int d = streamhost::dof_of(so.h_sidx[k]);
      const bool direct = so.h_price_skip[k] || (cnt2[d] == 1 && cnt[d] > 1);
      if (!direct) continue;
      const size_t e = k / P;
      const int m = (int)(k - e * P), t = m & 15, r = m >> 4;
      pp[(e * (npk + 1) + npk) * 16 + t] |= 2u << (2 * r);
      dropped += so.h_price_skip[k] ? 1 : 0, freed += so.h_price_skip[k] ? 0 : 1;
    }
    std::vector<int32_t> shared2;
    for (int d = 0; d < so.lsize; d++)
      if (cnt2[d] > 1) shared2.push_back(d);
    fprintf(stderr, "PALACE_AMD_PRICE_BLOCK=%d: %zu of %zu E-vector entries dropped, %zu more leave the gather; shared dofs %zu -> %zu\n",
            price_group, dropped, nnz, freed, so.h_shared.size(), shared2.size());
    so.h_shared = shared2;
    so.n_shared = (int)shared2.size();
  }
  if (so.fe_type == PA_FE_HCURL && !wide_form(so)) {
    // batches of four affine elements: kAffBit in the flag words of their 4 x 16 (element, lane) pairs -- the flag arrives with the
    // index words, one batch ahead of the q-data request it steers
    stream_affine_setup(so);
    const int npl = (P + 15) / 16, npk = (npl + 3) / 4;
    for (size_t b = 0; b < so.qd->batch_aff.size(); b++)
      if (so.qd->batch_aff[b])
        for (size_t e = 4 * b; e < 4 * b + 4; e++)
          for (int t = 0; t < 16; t++) pp[(e * (npk + 1) + npk) * 16 + t] |= streamhost::kAffBit;
  }
  so.h_perm_s = pp;
  if (so.fe_type == PA_FE_HCURL && !wide_form(so)) {
    // four-point H(curl) kernel: flag words on their own, slot words through the pattern dictionary (pa_internal.hpp)
    const int npl = (P + 15) / 16, npk = (npl + 3) / 4, nep = (ne + 3) & ~3, sw = npk * 16;
    std::vector<uint32_t> flagw((size_t)nep * 16), slots;
    std::map<std::vector<uint32_t>, uint32_t> dict;
    std::vector<uint32_t> key((size_t)sw);
    for (int e = 0; e < nep; e++) {
      const uint32_t *row = &pp[(size_t)e * (npk + 1) * 16];
      std::copy(row + sw, row + sw + 16, flagw.begin() + (size_t)e * 16);
      key.assign(row, row + sw);
      auto it = dict.find(key);
      if (it == dict.end()) {
        it = dict.emplace(key, (uint32_t)dict.size()).first;
        slots.insert(slots.end(), key.begin(), key.end());
      }
      ic[(size_t)e * streamhost::kIdxWords + streamhost::kIdxPattern] = it->second;
    }
    so.n_slot_patterns = (int)dict.size();
    so.d_flagw = dev_upload(flagw.data(), flagw.size());
    so.d_slots = dev_upload(slots.data(), slots.size());
  }
  so.d_idxc = dev_upload(ic.data(), ic.size());
  so.d_perm_s = dev_upload(pp.data(), pp.size());
  if (so.qd->metric || (so.iso && so.qf == PA_QF_HDIV_33 && so.geom->d_xnodes)) stream_element_coefficients(so);

  std::vector<uint32_t> code;
  std::vector<RunHdr> hdr;
  std::vector<int32_t> rpos;
  streamhost::build_runs(ne, P, so.lsize, so.h_sidx.data(), so.h_shared, code, hdr, rpos, nullptr,
                         so.h_price_skip.empty() ? nullptr : so.h_price_skip.data());
  {
    const std::vector<RunChunk> ch = streamhost::run_chunks(code);
    so.d_rchunk = dev_upload(reinterpret_cast<const uint32_t *>(ch.data()), 4 * ch.size());
  }
  so.d_rhdr = dev_upload(reinterpret_cast<const int32_t *>(hdr.data()), 2 * hdr.size());
  so.d_rpos = dev_upload(rpos.data(), rpos.size());
  so.n_runs = (int)hdr.size() - 1;
}

// Essential dofs (pa_op_set_essential): flagged (read as zero) and never exclusive in a copy of
// the flag words (their element-local result goes to the E-vector like a shared dof's), and present in a second run list,
// so that the run gather owns every essential row: it writes x or 0 there when ParOperator's fix-up is fused
// (rap.cpp:223-233) and the plain sum otherwise.
void stream_set_essential(SubOp &so, const std::vector<char> &flag) {
  if (!so.d_idxc) return;
  so.h_ess_flag = flag;
  if (so.n_all > 0) {  // (built for another list: again on the next pa_op_prepare_fused_step)
    hipFree(so.d_flagw_all), hipFree(so.d_perm_s_all), hipFree(so.d_rchunk_all), hipFree(so.d_rhdr_all), hipFree(so.d_rpos_all);
    so.d_flagw_all = so.d_perm_s_all = nullptr, so.d_rchunk_all = nullptr, so.d_rhdr_all = so.d_rpos_all = nullptr, so.n_all = 0;
  }
  const int P = so.P, npl = (P + 15) / 16, npk = (npl + 3) / 4;
  std::vector<uint32_t> pb(so.h_perm_s);
  const size_t nnz = (size_t)so.ne * so.P;
  const bool wide = wide_form(so);
  const int npkw = ((P + 31) / 32 + 1) / 2;
  for (size_t k = 0; k < nnz; k++) {
    if (flag[streamhost::dof_of(so.h_sidx[k])]) {
      const size_t e = k / P;
      if (wide) {  // flags ride in the slot half-words (pack_index_wide)
        const int m = (int)(k - e * P), t = m & 31, r = m >> 5;
        uint32_t &w = pb[(e * npkw + (r >> 1)) * 32 + t];
        w &= ~(streamhost::kWideExcl << (16 * (r & 1)));
        w |= streamhost::kWideEss << (16 * (r & 1));
        continue;
      }
      const int m = (int)(k - e * P), t = m & 15, r = m >> 4;
      uint32_t &fw = pb[(e * (npk + 1) + npk) * 16 + t];
      fw &= ~(2u << (2 * r));  // off the direct path
      fw |= 1u << (18 + r);    // read as zero
    }
  }
  hipFree(so.d_perm_s_bc);
  so.d_perm_s_bc = dev_upload(pb.data(), pb.size());
  if (so.d_flagw) {  // (the flag words of the masked copy; the slot words are the same)
    const int nep = (so.ne + 3) & ~3;
    std::vector<uint32_t> fb((size_t)nep * 16);
    for (int e = 0; e < nep; e++) std::copy(&pb[((size_t)e * (npk + 1) + npk) * 16], &pb[((size_t)e * (npk + 1) + npk) * 16] + 16, fb.begin() + (size_t)e * 16);
    hipFree(so.d_flagw_bc);
    so.d_flagw_bc = dev_upload(fb.data(), fb.size());
  }
  std::vector<int32_t> count((size_t)so.lsize, 0);
  for (size_t k = 0; k < nnz; k++) count[streamhost::dof_of(so.h_sidx[k])]++;
  std::vector<int32_t> shared;
  shared.reserve(so.h_shared.size());
  if (!so.h_price_skip.empty()) {  // (pricing experiment: copies taken off the E-vector are not copies)
    std::fill(count.begin(), count.end(), 0);
    for (size_t k = 0; k < nnz; k++)
      if (!so.h_price_skip[k]) count[streamhost::dof_of(so.h_sidx[k])]++;
  }
  for (int d = 0; d < so.lsize; d++)
    if (count[d] != 1 || flag[d]) shared.push_back(d);
  std::vector<uint32_t> code;
  std::vector<RunHdr> hdr;
  std::vector<int32_t> rpos;
  streamhost::build_runs(so.ne, P, so.lsize, so.h_sidx.data(), shared, code, hdr, rpos, flag.data(),  // (runs: all essential or none)
                         so.h_price_skip.empty() ? nullptr : so.h_price_skip.data());
  hipFree(so.d_rhdr_bc), hipFree(so.d_rpos_bc), hipFree(so.d_rchunk_bc);
  {
    const std::vector<RunChunk> ch = streamhost::run_chunks(code);
    so.d_rchunk_bc = dev_upload(reinterpret_cast<const uint32_t *>(ch.data()), 4 * ch.size());
  }
  so.d_rhdr_bc = dev_upload(reinterpret_cast<const int32_t *>(hdr.data()), 2 * hdr.size());
  so.d_rpos_bc = dev_upload(rpos.data(), rpos.size());
  so.n_shared_bc = (int)shared.size();
  so.n_runs_bc = (int)hdr.size() - 1;
}

// Interior / interface split for multi-rank applies: flag[d] != 0 marks the local dofs that take part in the halo exchange
// (ghosts and the owned dofs other ranks hold as ghosts).  Batches (four consecutive elements) without any such dof form
// list 0 and can run while the exchange is in flight; the others form list 1.
void stream_set_interface(SubOp &so, const std::vector<char> &flag) {
  if (!so.d_idxc) return;
  const int epb = wide_form(so) ? 2 : 4;  // elements per batch (one wave)
  const int nb = (so.ne + epb - 1) / epb, P = so.P;
  std::vector<int32_t> lists[2];
  for (int b = 0; b < nb; b++) {
    bool iface = false;
    for (int e = epb * b; e < std::min(epb * b + epb, so.ne) && !iface; e++)
      for (int m = 0; m < P && !iface; m++) iface = flag[streamhost::dof_of(so.h_sidx[(size_t)e * P + m])] != 0;
    lists[iface ? 1 : 0].push_back(b);
  }
  for (int ph = 0; ph < 2; ph++) {
    hipFree(so.d_blist[ph]);
    so.d_blist[ph] = lists[ph].empty() ? nullptr : dev_upload(lists[ph].data(), lists[ph].size());
    so.n_blist[ph] = (int)lists[ph].size();
  }
  so.has_blist = true;
}

void free_stream(SubOp &so) {
  hipFree(so.d_blist[0]), hipFree(so.d_blist[1]);
  hipFree(so.d_idxc), hipFree(so.d_perm_s), hipFree(so.d_perm_s_bc), hipFree(so.d_coef_s);
  hipFree(so.d_flagw), hipFree(so.d_flagw_bc), hipFree(so.d_slots);
  hipFree(so.d_flagw_all), hipFree(so.d_perm_s_all), hipFree(so.d_rchunk_all), hipFree(so.d_rhdr_all), hipFree(so.d_rpos_all);
  so.d_flagw_all = so.d_perm_s_all = nullptr, so.d_rchunk_all = nullptr, so.d_rhdr_all = so.d_rpos_all = nullptr, so.n_all = 0;
  hipFree(so.d_rhdr), hipFree(so.d_rpos), hipFree(so.d_rhdr_bc), hipFree(so.d_rpos_bc), hipFree(so.d_rchunk), hipFree(so.d_rchunk_bc);
}

static int device_cus() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}

template <int P1, bool U, bool C, bool METRIC, int MINW, int GPOS, bool CPLX = false, bool SPLIT = false, int GEOMN = 0>
static void launch_gpos(const SubOp &so, NDStreamArgs<P1> &a, hipStream_t s) {
  using L = typename std::conditional<P1 == 3 && !CPLX && C && !U, NDLayoutInPlaceSwz3, NDLayout<P1, 4>>::type;  // (as in the kernel)
  for (int i = 0; i < HalfTab<P1, 4>::LEN; i++) a.tab.Bo[i] = so.Bo[i];
  for (int i = 0; i < HalfTab<P1 + 1, 4>::LEN; i++) a.tab.Bc[i] = so.Bc[i], a.tab.Gc[i] = so.Gc[i];
  constexpr int PP = 3 * P1 * (P1 + 1) * (P1 + 1);
  constexpr int NPK = ((PP + 15) / 16 + 3) / 4;
  const size_t lds = sizeof(double) * (size_t)(kWavesPerBlock * 4) *
                     stream_lds_elem(L::ELEM_PAD + (PP + 1) / 2 + (NPK + 1) * 8 + 12 + (GEOMN ? 82 : 0));
  // resident workgroups per CU: registers (MINW waves per SIMD), LDS (160 KB), at most 8
  // workgroups per CU: what the registers (MINW waves per SIMD) and the LDS admit, and not more than the occupancy query
  // says -- with a fixed stride a workgroup that had to queue would run after the others and double the time
  static const int wg_env = getenv("PALACE_AMD_STREAM_WG") ? atoi(getenv("PALACE_AMD_STREAM_WG")) : 0;
  static const int per_cu_query = [&] {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nd_hex_stream_kernel<P1, U, C, METRIC, MINW, GPOS, CPLX, SPLIT, GEOMN>,
                                                     64 * kWavesPerBlock, lds) != hipSuccess || nb <= 0)
      nb = 8;
    return std::min({nb, MINW * 4 / kWavesPerBlock, (int)(160 * 1024 / lds), 8});
  }();
  const int per_cu = wg_env > 0 ? wg_env : per_cu_query;
  const int per_xcd = std::max(1, device_cus() / 8) * per_cu;
  if (CPLX) a.nbatch = (so.ne + 1) / 2;  // two elements times two parts per batch
  else if (!a.blist) a.nbatch = (so.ne + 3) / 4;  // (else: the length of the list, set by the caller)
  if (a.nbatch == 0) return;
  a.chunk = (a.nbatch + 7) / 8;
  // Few rounds (the per-rank size of a strong-scaling run: 490 batches per XCD over 384 resident waves = 1.28 rounds): the grid can
  // be sized so that every wave walks the SAME number of batches -- 245 waves x 2 instead of 106 x 2 + 278 x 1
  // (PALACE_AMD_STREAM_BALANCE=N: balance up to N rounds).  Measured neutral on the 1/8 slab (round 5, profiles/r05_halo_proxy.log:
  // local apply 29.0 against 29.1 us, ParOperator::Mult 45.3 both): the critical path is two batches either way.  Default off.
  static const int balance = getenv("PALACE_AMD_STREAM_BALANCE") ? atoi(getenv("PALACE_AMD_STREAM_BALANCE")) : 0;
  const int max_waves = per_xcd * kWavesPerBlock;
  int waves = std::min(max_waves, a.chunk);
  const int rounds = (a.chunk + max_waves - 1) / max_waves;
  if (rounds >= 2 && rounds <= balance) waves = (a.chunk + rounds - 1) / rounds;
  const int wgx = std::max(1, (waves + kWavesPerBlock - 1) / kWavesPerBlock);
  hipLaunchKernelGGL((nd_hex_stream_kernel<P1, U, C, METRIC, MINW, GPOS, CPLX, SPLIT, GEOMN>), dim3(8 * wgx), dim3(64 * kWavesPerBlock),
                     lds, s, a);
  PA_HIP(hipGetLastError());
}

// where x of the next batch is requested: after the first transposed component for the curl-curl kernel, after the second
// for the kernels with a mass term (their index words are requested just before D and need time to arrive and to be
// decoded; later = fewer live registers: 202 instead of 248 VGPRs for K + M at p = 3; measured on the 10M-dof case with the
// run-compressed index: curl-curl 0.180 / 0.181 ms, K + M 0.187 / 0.185 ms, mass 0.168 / 0.156 ms for positions 1 / 2).
// PALACE_AMD_STREAM_GPOS = 0 / 1 / 2 for A/B.
template <int P1, bool U, bool C, bool METRIC, int MINW_>
static void launch_variant(const SubOp &so, NDStreamArgs<P1> &a, hipStream_t s) {
#ifdef PA_STREAM_MINW2  // experiment builds: two waves per SIMD everywhere
  constexpr int MINW = 2;
#else
  constexpr int MINW = MINW_;
#endif
  static const int gpos = getenv("PALACE_AMD_STREAM_GPOS") ? atoi(getenv("PALACE_AMD_STREAM_GPOS")) : (U ? 2 : 1);
  if (a.nsplit >= 0)  // split vectors: the default gather position only (one more instantiation per operator kind)
    return launch_gpos<P1, U, C, METRIC, MINW, (U ? 2 : 1), false, true>(so, a, s);
  if (gpos == 0)
    launch_gpos<P1, U, C, METRIC, MINW, 0>(so, a, s);
  else if (gpos == 2)
    launch_gpos<P1, U, C, METRIC, MINW, 2>(so, a, s);
  else
    launch_gpos<P1, U, C, METRIC, MINW, 1>(so, a, s);
}

template <int P1>
static void launch_p(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase, const SplitIO *split,
                     bool all = false) {
  NDStreamArgs<P1> a{};  // (every field the chosen form does not use: zero)
  a.nsplit = -1, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr, a.yg = nullptr;
  if (split) {
    PA_REQUIRE(split->n_true >= 0 && split->n_true <= so.lsize, "split point outside the local vector");
    a.nsplit = split->n_true;
    a.xg0 = split->xg0 - split->n_true, a.xg1 = (split->xg1 ? split->xg1 : split->xg0) - split->n_true;
    a.xg_sel = split->sel, a.yg = split->yg - split->n_true;
  }
  a.ne = so.ne;
  a.blist = nullptr, a.nbatch = 0;
  if (phase >= 0) {  // 0: batches without interface elements, 1: the others (stream_set_interface)
    PA_REQUIRE(so.d_blist[phase] || so.n_blist[phase] == 0, "interface batch lists missing");
    a.blist = so.d_blist[phase], a.nbatch = so.n_blist[phase];
    if (a.nbatch == 0) return;
  }
  a.idxc = so.d_idxc;
  a.flagw = all ? so.d_flagw_all : (masked ? so.d_flagw_bc : so.d_flagw);
  a.slots = so.d_slots;
  a.qdata = so.qd->d;
  a.qaff = so.qd->d_aff, a.wq2[0] = so.qd->wq2[0], a.wq2[1] = so.qd->wq2[1];
#ifdef PA_METRIC6
  if (so.geom->w1.size() == 4) a.wq2[0] = so.geom->w1[0], a.wq2[1] = so.geom->w1[1];
#endif
  a.coef = so.d_coef_s;
  a.xn = nullptr, a.gtab = nullptr;
  a.x = x, a.y = y, a.ye = so.d_ye;
  const bool m = so.qd->metric;
  switch (so.qf) {
    case PA_QF_HDIV_33: {
      // geometry from the nodes (PALACE_AMD_STREAM_GEOM=nodes; order 3, isotropic coefficient, hex27 geometry, one vector): see the kernel
      // (read at every launch: the bench times both forms of one operator in one process)
      const char *ge = getenv("PALACE_AMD_STREAM_GEOM");
      const bool geomn = ge && std::string(ge) == "nodes";
      if (!m && geomn && P1 == 3 && so.iso && so.geom->d_xnodes && so.d_coef_s && !split) {
        a.xn = so.geom->d_xnodes, a.gtab = so.geom->d_gtab;
        // two waves per SIMD: the 27 partial sums are 54 more live registers in the D stage (222-244 in all; compiled for three
        // waves per SIMD the kernel spills 32-47 of them and runs at 284-331 us instead of 199: profiles/r05_geomn_first_form.log);
        // PALACE_AMD_GEOMN_VARIANT=w2g2 requests x of the next batch one transposed component later
        const char *ve = getenv("PALACE_AMD_GEOMN_VARIANT");
        if (ve && std::string(ve) == "w2g2") launch_gpos<P1, false, true, false, 2, 2, false, false, (P1 == 3 ? 1 : 0)>(so, a, s);
        else if (ve && std::string(ve) == "park3g1") launch_gpos<P1, false, true, false, 3, 1, false, false, (P1 == 3 ? 2 : 0)>(so, a, s);
        else if (ve && std::string(ve) == "park3g2") launch_gpos<P1, false, true, false, 3, 2, false, false, (P1 == 3 ? 2 : 0)>(so, a, s);
        else launch_gpos<P1, false, true, false, 2, 1, false, false, (P1 == 3 ? 1 : 0)>(so, a, s);
        break;
      }
      if (m) launch_variant<P1, false, true, true, (P1 == 3 ? 2 : 3)>(so, a, s); else launch_variant<P1, false, true, false, 3>(so, a, s);
    } break;
    case PA_QF_HCURL_33:
      if (m) launch_variant<P1, true, false, true, 3>(so, a, s); else launch_variant<P1, true, false, false, 3>(so, a, s);
      break;
    case PA_QF_HDIVMASS_33:
      // (packed D of both terms, anisotropic materials: twelve doubles per point, 96 registers of q-data per lane; round 5)
      if (m) launch_variant<P1, true, true, true, (P1 == 3 ? PA_KM_MINW : PA_KM_MINW_LOW)>(so, a, s);
      else launch_variant<P1, true, true, false, 2>(so, a, s);
      break;
    default: throw Error("QFunction not available for H(curl) hexahedra");
  }
}

void launch_nd_hex_stream_all(const SubOp &so, const double *x, hipStream_t s, const SplitIO *split) {
  PA_REQUIRE(so.n_all > 0, "stream_build_all has not been called");
  double *unused = so.d_ye;  // (no entry is exclusive in this form: y is never written)
  if (so.fe_type == PA_FE_H1) return launch_h1_hex_stream(so, x, unused, true, s, split, true);
  if (wide_form(so)) return launch_nd_hex_stream5(so, x, unused, true, s, -1, split, true);
  switch (so.p) {
    case 1: launch_p<1>(so, x, unused, true, s, -1, split, true); break;
    case 2: launch_p<2>(so, x, unused, true, s, -1, split, true); break;
    case 3: launch_p<3>(so, x, unused, true, s, -1, split, true); break;
    default: throw Error("no streaming H(curl) hex kernel for this order");
  }
}

bool nd_hex_stream_split_ok(const SubOp &so) {
  return so.fe_type == PA_FE_HCURL && so.d_idxc && (so.q1d == 4 || (wide_form(so) && nd_hex_stream5_ok(so)));
}

void launch_nd_hex_stream(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase, const SplitIO *split) {
  if (wide_form(so)) return launch_nd_hex_stream5(so, x, y, masked, s, phase, split);
  switch (so.p) {
    case 1: launch_p<1>(so, x, y, masked, s, phase, split); break;
    case 2: launch_p<2>(so, x, y, masked, s, phase, split); break;
    case 3: launch_p<3>(so, x, y, masked, s, phase, split); break;
    default: throw Error("no streaming H(curl) hex kernel for this order");
  }
}

// ---- complex form: y = (A_r + i A_i) x in one pass ---------------------------------------------------
// Both operators are metric-form (isotropic materials) H(curl) blocks on the same space and geometry with curl-curl and / or
// mass terms: they share the index arrays and the q-data and differ by their per-element scalar coefficients only.
bool nd_hex_stream_complex_ok(const SubOp &sr, const SubOp &si) {
  static const bool enabled = !(getenv("PALACE_AMD_COMPLEX_FUSED") && atoi(getenv("PALACE_AMD_COMPLEX_FUSED")) == 0);
  auto qf_ok = [](const SubOp &so) {
    return so.fe_type == PA_FE_HCURL && (so.qf == PA_QF_HDIV_33 || so.qf == PA_QF_HCURL_33 || so.qf == PA_QF_HDIVMASS_33);
  };
  if (!enabled || !qf_ok(sr) || !qf_ok(si)) return false;
  if (sr.geom != si.geom || sr.ne != si.ne || sr.p != si.p || sr.P != si.P || sr.q1d != si.q1d) return false;
  // metric form (isotropic materials): the real operator provides the kernel's arrays (the q-data is the geometry's J^T J, shared by
  // every such operator on it); the imaginary one only its per-element scalars (stream_element_coefficients)
  auto iso_ok = [](const SubOp &so) {
    return so.iso && ((so.q1d == 4 && so.p <= 3) || (so.q1d == 5 && so.p <= 4 && nd_hex_stream5_ok(so))) && !so.geom->h_attr.empty();
  };
  const bool metric = iso_ok(sr) && iso_ok(si) && sr.d_idxc && sr.d_coef_s && sr.qd && sr.qd->metric;
  // packed form (anisotropic materials, four points per direction; round 5): each operator's own packed symmetric D, 6 or 12 per point
  auto packed_ok = [](const SubOp &so) {
    return so.q1d == 4 && so.p <= 3 && so.qd && !so.qd->metric && so.qd->ncomp == (so.qf == PA_QF_HDIVMASS_33 ? 12 : 6);
  };
  const bool packed = packed_ok(sr) && packed_ok(si) && sr.d_idxc;
  if (!metric && !packed) return false;
  return sr.h_sidx == si.h_sidx;  // same restriction (host compare; the callers cache the answer)
}

template <int P1>
static void launch_complex_p(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                             double *ye_i, bool masked, hipStream_t s) {
  NDStreamArgs<P1> a{};  // (every field the chosen form does not use: zero)
  a.ne = sr.ne, a.blist = nullptr, a.nbatch = 0;
  a.idxc = sr.d_idxc;
  a.flagw = masked ? sr.d_flagw_bc : sr.d_flagw;
  a.slots = sr.d_slots;
  a.qdata = sr.qd->d;
  if (sr.qd->metric) a.qaff = sr.qd->d_aff, a.wq2[0] = sr.qd->wq2[0], a.wq2[1] = sr.qd->wq2[1];  // (affine batches: the metric form)
  a.coef = sr.d_coef_s, a.coef1 = si.d_coef_s;
  a.xn = nullptr, a.gtab = nullptr;
  a.x = xr, a.x1 = xi, a.y = yr, a.y1 = yi, a.ye = sr.d_ye, a.ye1 = ye_i;
  a.nsplit = -1, a.xg0 = a.xg1 = nullptr, a.xg_sel = nullptr, a.yg = nullptr;
  a.qdata1 = nullptr;
  if (sr.qd->metric) return launch_gpos<P1, true, true, true, 2, 2, true>(sr, a, s);
  const SubOp *ops[2] = {&sr, &si};
  for (int g = 0; g < 2; g++) {  // (pa_nd_hex.hip: packed q-data holds the mass block first, then the curl-curl block)
    const int qf = ops[g]->qf;
    a.qn[g] = ops[g]->qd->ncomp;
    a.qm[g] = (qf == PA_QF_HCURL_33 || qf == PA_QF_HDIVMASS_33) ? 0 : -1;
    a.qc[g] = qf == PA_QF_HDIV_33 ? 0 : (qf == PA_QF_HDIVMASS_33 ? 6 : -1);
  }
  a.qdata1 = si.qd->d;
  // (where x of the next batch is requested: after the second transposed component, as in the real K + M kernel;
  // PALACE_AMD_CPLX_GPOS=1 / 3 for A / B, read at every launch)
  const char *ge = getenv("PALACE_AMD_CPLX_GPOS");
  const int gpos = ge ? atoi(ge) : 2;
  if (gpos == 1) launch_gpos<P1, true, true, false, 2, 1, true>(sr, a, s);
  else if (gpos == 3) launch_gpos<P1, true, true, false, 2, 3, true>(sr, a, s);
  else launch_gpos<P1, true, true, false, 2, 2, true>(sr, a, s);
}

void launch_nd_hex_stream_complex(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                                  double *ye_i, bool masked, hipStream_t s) {
  if (sr.q1d == 5) return launch_nd_hex_stream5_complex(sr, si, xr, xi, yr, yi, ye_i, masked, s);
  switch (sr.p) {
    case 1: launch_complex_p<1>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    case 2: launch_complex_p<2>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    case 3: launch_complex_p<3>(sr, si, xr, xi, yr, yi, ye_i, masked, s); break;
    default: throw Error("no streaming H(curl) hex kernel for this order");
  }
}

// masked: the run list that owns the essential rows (the element kernel then ran on the _bc index arrays); ess_policy >= 0
// additionally fuses ParOperator's fix-up y[ess] = x[ess] | 0 into it
void launch_et_run_gather(const SubOp &so, double *y, bool accumulate, hipStream_t s, const double *x, bool masked,
                          int ess_policy, const double *ye, const SplitIO *split) {
  const int n = masked ? so.n_shared_bc : so.n_shared;
  if (n == 0) return;
  PA_REQUIRE(!split || !accumulate, "split vectors: y = A x only");
  hipLaunchKernelGGL(et_run_gather_kernel_t<false>, dim3((n + 256 * kGatherILP - 1) / (256 * kGatherILP)), dim3(256), 0, s, n,
                     reinterpret_cast<const RunChunk *>(masked ? so.d_rchunk_bc : so.d_rchunk),
                     reinterpret_cast<const RunHdr *>(masked ? so.d_rhdr_bc : so.d_rhdr),
                     masked ? so.d_rpos_bc : so.d_rpos, ye ? ye : so.d_ye, y, accumulate ? 1 : 0, x, masked ? ess_policy : -1,
                     split ? split->n_true : 0x7fffffff, split ? split->yg - split->n_true : nullptr, GatherStep{}, GatherPart2{});
  PA_HIP(hipGetLastError());
}

// the two parts of a one-pass complex apply in one launch: y = sum of `so.d_ye`, y1 = sum of ye1 (same runs, same order)
void launch_et_run_gather2(const SubOp &so, double *y, double *y1, hipStream_t s, const double *x, const double *x1, bool masked,
                           int ess_policy, const double *ye1) {
  const int n = masked ? so.n_shared_bc : so.n_shared;
  if (n == 0) return;
  hipLaunchKernelGGL((et_run_gather_kernel_t<false, true>), dim3((n + 256 * kGatherILP - 1) / (256 * kGatherILP)), dim3(256), 0, s, n,
                     reinterpret_cast<const RunChunk *>(masked ? so.d_rchunk_bc : so.d_rchunk),
                     reinterpret_cast<const RunHdr *>(masked ? so.d_rhdr_bc : so.d_rhdr),
                     masked ? so.d_rpos_bc : so.d_rpos, so.d_ye, y, 0, x, masked ? ess_policy : -1, 0x7fffffff, nullptr, GatherStep{},
                     GatherPart2{ye1, x1, y1});
  PA_HIP(hipGetLastError());
}

// ---- the fused smoother step: every dof through the E-vector, consumed by the gather's epilogue -------------------------------
bool stream_build_all(SubOp &so) {
  if (so.n_all > 0) return true;
  const bool wide = wide_form(so), h1 = so.fe_type == PA_FE_H1;
  // (H1 blocks on the streaming kernel: the flag word is the last row of the element's block of slot words, pa_h1_hex_stream.hip)
  if (!so.d_idxc || (!h1 && so.fe_type != PA_FE_HCURL) || (!wide && !h1 && !so.d_flagw) || so.h_perm_s.empty()) return false;
  const int P = so.P, npl = (P + 15) / 16, npk = (npl + 3) / 4, nep = (so.ne + 3) & ~3;
  const size_t nnz = (size_t)so.ne * P;
  std::vector<char> flag(so.h_ess_flag);
  flag.resize((size_t)so.lsize, 0);
  std::vector<uint32_t> fw;  // four-point kernel: the flag words; five-point kernel: the slot half-words with their flags
  const int npkw = ((P + 31) / 32 + 1) / 2;
  if (wide) {
    fw = so.h_perm_s;
    for (uint32_t &w : fw) w &= ~(streamhost::kWideExcl | (streamhost::kWideExcl << 16));  // nothing takes the direct path
  } else if (h1) {
    fw = so.h_perm_s;  // [nep][npk + 1][16]: the flag words are row npk
    for (int e = 0; e < nep; e++)
      for (int t = 0; t < 16; t++)
        for (int r = 0; r < npl; r++) fw[((size_t)e * (npk + 1) + npk) * 16 + t] &= ~(2u << (2 * r));
  } else {
    fw.resize((size_t)nep * 16);
    for (int e = 0; e < nep; e++)
      for (int t = 0; t < 16; t++) {
        uint32_t w = so.h_perm_s[((size_t)e * (npk + 1) + npk) * 16 + t];
        for (int r = 0; r < npl; r++) w &= ~(2u << (2 * r));  // nothing takes the direct path
        fw[(size_t)e * 16 + t] = w;
      }
  }
  std::vector<char> present((size_t)so.lsize, 0);
  for (size_t k = 0; k < nnz; k++) {
    const int d = streamhost::dof_of(so.h_sidx[k]);
    present[d] = 1;
    if (flag[d]) {
      const size_t e = k / P;
      if (wide) {
        const int m = (int)(k - e * P), t = m & 31, r = m >> 5;
        fw[(e * npkw + (r >> 1)) * 32 + t] |= streamhost::kWideEss << (16 * (r & 1));  // read as zero
      } else if (h1) {
        const int m = (int)(k - e * P), t = m & 15, r = m >> 4;
        fw[(e * (npk + 1) + npk) * 16 + t] |= 1u << (18 + r);
      } else {
        const int m = (int)(k - e * P), t = m & 15, r = m >> 4;
        fw[e * 16 + t] |= 1u << (18 + r);  // read as zero
      }
    }
  }
  std::vector<int32_t> all;
  all.reserve((size_t)so.lsize);
  for (int d = 0; d < so.lsize; d++)
    if (present[d] || flag[d]) all.push_back(d);
  if ((int)all.size() != so.lsize) return false;  // (a dof no element holds: the plain forms leave it alone, the step must not)
  std::vector<uint32_t> code;
  std::vector<RunHdr> hdr;
  std::vector<int32_t> rpos;
  streamhost::build_runs(so.ne, P, so.lsize, so.h_sidx.data(), all, code, hdr, rpos, flag.data(), nullptr);
  const std::vector<RunChunk> ch = streamhost::run_chunks(code);
  so.d_rchunk_all = dev_upload(reinterpret_cast<const uint32_t *>(ch.data()), 4 * ch.size());
  so.d_rhdr_all = dev_upload(reinterpret_cast<const int32_t *>(hdr.data()), 2 * hdr.size());
  so.d_rpos_all = dev_upload(rpos.data(), rpos.size());
  so.n_all = (int)all.size(), so.n_runs_all = (int)hdr.size() - 1;
  if (wide || h1)
    so.d_perm_s_all = dev_upload(fw.data(), fw.size());
  else
    so.d_flagw_all = dev_upload(fw.data(), fw.size());
  return true;
}

void launch_et_run_gather_step(const SubOp &so, const double *x, const GatherStep &step, int ess_policy, hipStream_t s,
                               const SplitIO *split) {
  PA_REQUIRE(so.n_all > 0, "stream_build_all has not been called");
  PA_REQUIRE(!split || (step.iface_mask && step.t_iface), "split form of the fused step: interface mask and buffer missing");
  const int n = so.n_all;
  hipLaunchKernelGGL(et_run_gather_kernel_t<true>, dim3((n + 256 * kGatherILP - 1) / (256 * kGatherILP)), dim3(256), 0, s, n,
                     reinterpret_cast<const RunChunk *>(so.d_rchunk_all), reinterpret_cast<const RunHdr *>(so.d_rhdr_all), so.d_rpos_all,
                     so.d_ye, nullptr, 0, x, ess_policy, split ? split->n_true : 0x7fffffff, split ? split->yg - split->n_true : nullptr,
                     step, GatherPart2{});
  PA_HIP(hipGetLastError());
}

#ifdef PA_STREAM_TRACE
}  // namespace pa
extern "C" int pa_debug_stream_trace(unsigned long long *out, int n) {
  const int total = pa::kTraceWG * pa::kTraceBatches * pa::kTraceStamps;
  if (n < total) return -total;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pa::g_trace), sizeof(unsigned long long) * total) != hipSuccess) return -1;
  return total;
}
namespace pa {
#endif
}  // namespace pa
