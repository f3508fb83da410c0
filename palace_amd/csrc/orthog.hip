// Gram-Schmidt of one Krylov column against the basis with the HOST OUT OF THE LOOP
// (reference: linalg/orthog.hpp:41-89 OrthogonalizeColumnMGS / CGS, used by iterative.cpp:629-633 and :820-824, followed there by
// H(j+1, j) = Norml2(w); w *= 1 / H(j+1, j)).
//
// The reference's arithmetic is kept (same inner products, same updates, in the same order); what changes is where the scalars
// live between the kernels:
//   * the inner products of a pass are reduced ON THE DEVICE (per-block partial sums, the last block to finish adds them in
//     block order -- a fixed order, so results are reproducible run to run) into a coefficient array in device memory, summed
//     over the ranks there (Comm::AllReduceSum on the stream), and the update kernels READ THEM FROM THERE;
//   * classical Gram-Schmidt (CGS, CGS2 = one refinement pass) takes kGB basis vectors per pass over w: 2 m / kGB launches per
//     pass instead of 3 m launches and m host synchronisations;
//   * modified Gram-Schmidt chains "w -= h_j v_j" with the inner product of the NEW w and v_{j+1} in one kernel: 4 vector
//     passes and one launch per basis vector instead of 5 passes, three launches and one synchronisation;
//   * the norm of the result is accumulated by the last update kernel and the normalisation kernel reads it from the device.
// One column costs ONE host synchronisation (the copy of the coefficients and the norm, which the Givens recursion needs on
// the host) whatever m is.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

#include "comm.hpp"
#include "complex.hpp"

namespace palace {

namespace {

constexpr int kGB = 8;  // basis vectors per classical pass
constexpr int kBlk = 256, kMaxBlk = 1024;

struct BasisPtrs {
  const double *r[kGB], *i[kGB];
};

template <int W>
struct LaneT {
  using type = double;
};
template <>
struct LaneT<2> {
  using type = double2;
};
__device__ __forceinline__ double mul_acc(double acc, double a, double b) { return acc + a * b; }
__device__ __forceinline__ double2 mul_acc(double2 acc, double2 a, double2 b) { return {acc.x + a.x * b.x, acc.y + a.y * b.y}; }
__device__ __forceinline__ double mul_sub(double acc, double a, double b) { return acc - a * b; }
__device__ __forceinline__ double2 mul_sub(double2 acc, double2 a, double2 b) { return {acc.x - a.x * b.x, acc.y - a.y * b.y}; }
__device__ __forceinline__ double hsum(double a) { return a; }
__device__ __forceinline__ double hsum(double2 a) { return a.x + a.y; }
__device__ __forceinline__ double bcast(double, double s) { return s; }
__device__ __forceinline__ double2 bcast(double2, double s) { return {s, s}; }
template <class T>
__device__ __forceinline__ T ld(const double *p, long long i) {
  return reinterpret_cast<const T *>(p)[i];
}
template <class T>
__device__ __forceinline__ void st(double *p, long long i, T v) {
  reinterpret_cast<T *>(p)[i] = v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Per-block sums of NV per-thread values -> partial[k * gridDim.x + blockIdx.x]; the block that finishes last adds the partial
// sums of every block in block order and writes out[k].  `counter` returns to zero for the next launch.
template <int NV>
__device__ __forceinline__ void grid_reduce(double (&v)[NV], int nv, double *__restrict__ partial, unsigned *__restrict__ counter,
                                            double *__restrict__ out) {
  __shared__ double sm[NV][kBlk / 64];
  __shared__ unsigned ticket;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const double s = wave_sum(v[k]);
    if (lane == 0) sm[k][wave] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < nv) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < kBlk / 64; q++) s += sm[threadIdx.x][q];
#if defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__)
    // ARCHITECTURE ASSUMPTION (gfx9 family: one in-order counter, vmcnt, covers loads AND stores, and agent-scope stores are
    // written through to the level every XCD sees): waiting for the relaxed store to be performed orders it before the ticket
    // below WITHOUT a release fence -- a fence writes the whole L2 back, once per block: measured 100 us per launch.  The
    // run-to-run bit-reproducibility test (tests/test_orthog_gpu.py) is the gate.
    __hip_atomic_store(&partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    // any other target (separate store counter, other cache policies): the HIP memory model's release / acquire pair
    __hip_atomic_store(&partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
  }
  __syncthreads();
#if defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__)
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#endif
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
#if !(defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__))
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  for (int k = 0; k < nv; k++) {
    double s = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kBlk)
      s += __hip_atomic_load(&partial[(size_t)k * gridDim.x + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) sm[0][wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < kBlk / 64; q++) t += sm[0][q];
      out[k] = t;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[NC j + {0, 1}] = (x, v_j) = v_j^H x for the m <= kGB vectors of the batch (vector.cpp:674-685: Dot(x, y) = y^H x)
template <bool CPLX, int W>
__global__ __launch_bounds__(kBlk) void k_gs_dot(const double *__restrict__ xr, const double *__restrict__ xi, const BasisPtrs V,
                                                 const int m, const long long n, double *__restrict__ partial,
                                                 unsigned *__restrict__ counter, double *__restrict__ out) {
  using T = typename LaneT<W>::type;
  constexpr int NC = CPLX ? 2 : 1;
  T acc[kGB * NC];
#pragma unroll
  for (int k = 0; k < kGB * NC; k++) acc[k] = T{};
  const long long nv = n / W;
  for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < nv; i += (long long)gridDim.x * kBlk) {
    const T a = ld<T>(xr, i), b = CPLX ? ld<T>(xi, i) : T{};
#pragma unroll
    for (int j = 0; j < kGB; j++)
      if (j < m) {
        const T c = ld<T>(V.r[j], i);
        acc[NC * j] = mul_acc(acc[NC * j], a, c);
        if (CPLX) {
          const T d = ld<T>(V.i[j], i);
          acc[NC * j] = mul_acc(acc[NC * j], b, d);
          acc[NC * j + 1] = mul_sub(mul_acc(acc[NC * j + 1], b, c), a, d);
        }
      }
  }
  double v[kGB * NC];
#pragma unroll
  for (int k = 0; k < kGB * NC; k++) v[k] = hsum(acc[k]);
  if (W > 1 && (n % W) && blockIdx.x == 0 && threadIdx.x == 0) {  // the odd last entry
    const long long i = n - 1;
    const double a = xr[i], b = CPLX ? xi[i] : 0.0;
#pragma unroll
    for (int j = 0; j < kGB; j++)
      if (j < m) {
        const double c = V.r[j][i];
        v[NC * j] += a * c;
        if (CPLX) {
          const double d = V.i[j][i];
          v[NC * j] += b * d;
          v[NC * j + 1] += b * c - a * d;
        }
      }
  }
  grid_reduce<kGB * NC>(v, m * NC, partial, counter, out);
}

// w -= sum_{j < m} h_j v_j with h read from DEVICE memory (coef[NC j + {0, 1}]), then, on the updated w:
//   TAIL_NORM: out[0] = ||w||^2          TAIL_DOT: out[0 .. NC) = (w, next)          TAIL_NONE: nothing
enum { TAIL_NONE = 0, TAIL_NORM = 1, TAIL_DOT = 2 };
template <bool CPLX, int W, int TAIL>
__global__ __launch_bounds__(kBlk) void k_gs_update(double *__restrict__ wr, double *__restrict__ wi, const BasisPtrs V, const int m,
                                                    const double *__restrict__ coef, const double *__restrict__ nr,
                                                    const double *__restrict__ ni, const long long n,
                                                    double *__restrict__ partial, unsigned *__restrict__ counter,
                                                    double *__restrict__ out) {
  using T = typename LaneT<W>::type;
  constexpr int NC = CPLX ? 2 : 1;
  double hr[kGB], hi[kGB];
#pragma unroll
  for (int j = 0; j < kGB; j++) {
    hr[j] = j < m ? coef[NC * j] : 0.0;
    hi[j] = (CPLX && j < m) ? coef[NC * j + 1] : 0.0;
  }
  T t0 = T{}, t1 = T{};
  const long long nv = n / W;
  auto body = [&](auto tag, const long long i, auto &s0, auto &s1) {
    using U = decltype(tag);
    U a = ld<U>(wr, i), b = CPLX ? ld<U>(wi, i) : U{};
#pragma unroll
    for (int j = 0; j < kGB; j++)
      if (j < m) {
        const U c = ld<U>(V.r[j], i);
        a = mul_sub(a, bcast(U{}, hr[j]), c);  // (hr + i hi)(c + i d) = (hr c - hi d) + i (hr d + hi c)
        if (CPLX) {
          const U d = ld<U>(V.i[j], i);
          a = mul_acc(a, bcast(U{}, hi[j]), d);
          b = mul_sub(mul_sub(b, bcast(U{}, hr[j]), d), bcast(U{}, hi[j]), c);
        }
      }
    if (m > 0) {
      st<U>(wr, i, a);
      if (CPLX) st<U>(wi, i, b);
    }
    if (TAIL == TAIL_NORM) {
      s0 = mul_acc(s0, a, a);
      if (CPLX) s0 = mul_acc(s0, b, b);
    } else if (TAIL == TAIL_DOT) {
      const U c = ld<U>(nr, i);
      s0 = mul_acc(s0, a, c);
      if (CPLX) {
        const U d = ld<U>(ni, i);
        s0 = mul_acc(s0, b, d);
        s1 = mul_sub(mul_acc(s1, b, c), a, d);
      }
    }
  };
  for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < nv; i += (long long)gridDim.x * kBlk) body(T{}, i, t0, t1);
  double v[2] = {hsum(t0), hsum(t1)};
  if (W > 1 && (n % W) && blockIdx.x == 0 && threadIdx.x == 0) body(double{}, n - 1, v[0], v[1]);
  if (TAIL != TAIL_NONE) grid_reduce<2>(v, TAIL == TAIL_DOT ? NC : 1, partial, counter, out);
}

// w *= 1 / sqrt(|nrm2|) with the squared norm read from the device (iterative.cpp:632-633)
template <bool CPLX>
__global__ __launch_bounds__(kBlk) void k_gs_scale(double *__restrict__ wr, double *__restrict__ wi, const long long n,
                                                   const double *__restrict__ nrm2) {
  const double s = 1.0 / sqrt(fabs(nrm2[0]));
  for (long long i = (long long)blockIdx.x * kBlk + threadIdx.x; i < n; i += (long long)gridDim.x * kBlk) {
    wr[i] *= s;
    if (CPLX) wi[i] *= s;
  }
}

inline int grid_for(long long n) { return (int)std::max(1LL, std::min<long long>((n + kBlk - 1) / kBlk, kMaxBlk)); }
inline uintptr_t bits(const void *p) { return reinterpret_cast<uintptr_t>(p); }

struct Column {  // the vectors of one column as raw pointers (imaginary parts null for real scalars)
  const double *const *vr, *const *vi;
  double *wr, *wi;
  const double *xr, *xi;  // what the inner products are taken of: w itself, or W w for a weighted inner product
  long long n;
  int m;
};

// device scratch: [counter | partial sums 2 kGB kMaxBlk | coefficients pass 1 (NC m) | pass 2 (NC m) | ||w||^2]
struct GsBuffers {
  unsigned *counter;
  double *partial, *coef1, *coef2, *nrm2, *host;
};
GsBuffers buffers(const Context &c, int m) {
  Workspace &w = c.Work();
  const size_t head = 2 + (size_t)2 * kGB * kMaxBlk;
  double *d = w.GsDevice(head + 4 * (size_t)m + 2);
  return {reinterpret_cast<unsigned *>(d), d + 2, d + head, d + head + 2 * (size_t)m, d + head + 4 * (size_t)m,
          w.GsPinned(4 * (size_t)m + 2)};  // (m >= 1 here)
}

template <bool CPLX>
void run_column(const Context &c, Orthogonalization kind, const Column &col, bool normalize, double *H, double *hn) {
  constexpr int NC = CPLX ? 2 : 1;
  StreamGraph::RequireNotRecording("linalg::OrthogonalizeColumn");
  const int m = col.m, mm = std::max(col.m, 1);
  const GsBuffers B = buffers(c, mm);
  const bool weighted = col.xr != col.wr;
  uintptr_t al = bits(col.wr) | bits(col.wi) | bits(col.xr) | bits(col.xi);
  for (int j = 0; j < m; j++) al |= bits(col.vr[j]) | (CPLX ? bits(col.vi[j]) : 0);
  const bool wide = (al & 15) == 0 && col.n >= 2;
  const int nb = grid_for(wide ? (col.n + 1) / 2 : col.n);
  auto ptrs = [&](int j0, int mb) {
    BasisPtrs P{};
    for (int j = 0; j < mb; j++) P.r[j] = col.vr[j0 + j], P.i[j] = CPLX ? col.vi[j0 + j] : nullptr;
    return P;
  };
  auto dots = [&](double *coef) {  // one classical pass of inner products, all from the same x
    for (int j0 = 0; j0 < m; j0 += kGB) {
      const int mb = std::min(kGB, m - j0);
      if (wide)
        hipLaunchKernelGGL((k_gs_dot<CPLX, 2>), dim3(nb), dim3(kBlk), 0, c.stream, col.xr, col.xi, ptrs(j0, mb), mb, col.n, B.partial,
                           B.counter, coef + NC * j0);
      else
        hipLaunchKernelGGL((k_gs_dot<CPLX, 1>), dim3(nb), dim3(kBlk), 0, c.stream, col.xr, col.xi, ptrs(j0, mb), mb, col.n, B.partial,
                           B.counter, coef + NC * j0);
    }
    PA_HIP(hipGetLastError());
    if (c.comm) c.comm->AllReduceSum(coef, NC * m, c.stream);  // Mpi::GlobalSum of the whole column (orthog.hpp:73-76)
  };
  auto update = [&](int j0, int mb, const double *coef, int tail, const double *nr, const double *ni, double *out) {
#define PA_GS_UPDATE(W, TAIL)                                                                                                    \
  hipLaunchKernelGGL((k_gs_update<CPLX, W, TAIL>), dim3(nb), dim3(kBlk), 0, c.stream, col.wr, col.wi, ptrs(j0, mb), mb, coef, nr, ni, \
                     col.n, B.partial, B.counter, out)
    if (wide) {
      if (tail == TAIL_NONE) PA_GS_UPDATE(2, TAIL_NONE);
      else if (tail == TAIL_NORM) PA_GS_UPDATE(2, TAIL_NORM);
      else PA_GS_UPDATE(2, TAIL_DOT);
    } else {
      if (tail == TAIL_NONE) PA_GS_UPDATE(1, TAIL_NONE);
      else if (tail == TAIL_NORM) PA_GS_UPDATE(1, TAIL_NORM);
      else PA_GS_UPDATE(1, TAIL_DOT);
    }
#undef PA_GS_UPDATE
  };
  auto updates = [&](const double *coef, bool norm_at_end) {
    for (int j0 = 0; j0 < m; j0 += kGB) {
      const int mb = std::min(kGB, m - j0);
      update(j0, mb, coef + NC * j0, (norm_at_end && j0 + kGB >= m) ? TAIL_NORM : TAIL_NONE, nullptr, nullptr, B.nrm2);
    }
    PA_HIP(hipGetLastError());
  };

  bool two_passes = false;
  if (m == 0) {
    if (normalize) update(0, 0, B.coef1, TAIL_NORM, nullptr, nullptr, B.nrm2);
  } else if (kind == Orthogonalization::MGS) {
    PA_REQUIRE(!weighted, "the device-chained modified Gram-Schmidt takes the plain inner product");
    update(0, 0, B.coef1, TAIL_DOT, col.vr[0], CPLX ? col.vi[0] : nullptr, B.coef1);  // (w, v_0)
    for (int j = 0; j < m; j++) {
      if (c.comm) c.comm->AllReduceSum(B.coef1 + NC * j, NC, c.stream);
      if (j + 1 == m)
        update(j, 1, B.coef1 + NC * j, normalize ? TAIL_NORM : TAIL_NONE, nullptr, nullptr, B.nrm2);
      else  // w -= h_j v_j and (w, v_{j+1}) of the new w in the same pass
        update(j, 1, B.coef1 + NC * j, TAIL_DOT, col.vr[j + 1], CPLX ? col.vi[j + 1] : nullptr, B.coef1 + NC * (j + 1));
    }
    PA_HIP(hipGetLastError());
  } else {
    dots(B.coef1);
    updates(B.coef1, normalize && kind == Orthogonalization::CGS);
    if (kind == Orthogonalization::CGS2) {
      PA_REQUIRE(!weighted, "the refinement pass of a weighted inner product is driven by the caller");
      dots(B.coef2);
      updates(B.coef2, normalize);
      two_passes = true;
    }
  }
  if (normalize) {
    if (c.comm) c.comm->AllReduceSum(B.nrm2, 1, c.stream);
    hipLaunchKernelGGL((k_gs_scale<CPLX>), dim3(grid_for(col.n)), dim3(kBlk), 0, c.stream, col.wr, col.wi, col.n, B.nrm2);
    PA_HIP(hipGetLastError());
  }
  // the column of the Hessenberg matrix goes to the host in ONE copy: [coef1 (2 mm) | coef2 (2 mm) | nrm2] is contiguous
  PA_HIP(hipMemcpyAsync(B.host, B.coef1, sizeof(double) * (4 * (size_t)mm + 1), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
  if (c.comm) c.comm->PeerCheckNow();  // (a timed-out wait of the peer transport surfaces here, not as a wrong sum)
  for (int k = 0; k < NC * m; k++) H[k] = two_passes ? B.host[k] + B.host[2 * (size_t)mm + k] : B.host[k];
  if (normalize) *hn = std::sqrt(std::fabs(B.host[4 * (size_t)mm]));
}

int &gs_mode() {  // 1 = coefficients on the device (default), 0 = the host drives every inner product (PALACE_AMD_GS=host)
  static int mode = [] {
    const char *e = std::getenv("PALACE_AMD_GS");
    return (e && std::string(e) == "host") ? 0 : 1;
  }();
  return mode;
}
bool device_gs() { return gs_mode() != 0; }

}  // namespace

namespace linalg {

bool DeviceOrthogonalization() { return device_gs(); }
void SetDeviceOrthogonalization(bool on) { gs_mode() = on ? 1 : 0; }

void OrthogonalizeColumnDevice(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, const Vector *x,
                               double *H, int m, bool normalize, double *hn) {
  std::vector<const double *> vr((size_t)m);
  for (int j = 0; j < m; j++) vr[j] = V[j].Data();
  const Column col{vr.data(), nullptr, w.Data(), nullptr, x ? x->Data() : w.Data(), nullptr, w.Size(), m};
  run_column<false>(c, kind, col, normalize, H, hn);
}
void OrthogonalizeColumnDevice(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                               const ComplexVector *x, std::complex<double> *H, int m, bool normalize, double *hn) {
  std::vector<const double *> vr((size_t)m), vi((size_t)m);
  for (int j = 0; j < m; j++) vr[j] = V[j].Real().Data(), vi[j] = V[j].Imag().Data();
  const Column col{vr.data(),          vi.data(), w.Real().Data(), w.Imag().Data(), x ? x->Real().Data() : w.Real().Data(),
                   x ? x->Imag().Data() : w.Imag().Data(), w.Size(), m};
  run_column<true>(c, kind, col, normalize, reinterpret_cast<double *>(H), hn);  // std::complex<double> is two doubles (re, im)
}

double OrthonormalizeColumn(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, double *H, int m) {
  PA_REQUIRE(m >= 0 && (size_t)m <= V.size(), "Out of bounds number of columns for orthogonalization!");
  for (int j = 0; j < m; j++) PA_REQUIRE(V[j].Size() == w.Size(), "size mismatch in OrthonormalizeColumn");
  double hn = 0.0;
  if (device_gs()) {
    OrthogonalizeColumnDevice(c, kind, V, w, nullptr, H, m, true, &hn);
  } else {  // the reference's three statements (iterative.cpp:629-633), one call each
    OrthogonalizeColumn(c, kind, V, w, H, m);
    hn = Norml2(c, w);
    Scale(c, 1.0 / hn, w);
  }
  return hn;
}
double OrthonormalizeColumn(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                            std::complex<double> *H, int m) {
  PA_REQUIRE(m >= 0 && (size_t)m <= V.size(), "Out of bounds number of columns for orthogonalization!");
  for (int j = 0; j < m; j++) PA_REQUIRE(V[j].Size() == w.Size(), "size mismatch in OrthonormalizeColumn");
  double hn = 0.0;
  if (device_gs()) {
    OrthogonalizeColumnDevice(c, kind, V, w, nullptr, H, m, true, &hn);
  } else {
    OrthogonalizeColumn(c, kind, V, w, H, m);
    hn = Norml2(c, w);
    Scale(c, 1.0 / hn, w);
  }
  return hn;
}

}  // namespace linalg
}  // namespace palace
