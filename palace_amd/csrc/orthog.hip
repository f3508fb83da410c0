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

// ---- modified Gram-Schmidt with w RESIDENT IN THE REGISTER FILE (round 6) ---------------------------------------------------------
// The chained form above passes over w once per basis vector (read w, v_j, v_{j+1}, write w: four vector passes per column of
// H).  MI355X has 128 MB of vector registers: a Krylov vector of a few million entries fits in them, spread over one block per
// compute unit.  This kernel loads w ONCE, keeps it in registers for the whole column, reads every v_j once (its slice stays in
// registers between the inner product and the update), and stores w once -- normalised.  The inner product of step j is a grid-wide
// sum: per-block partial sums, a grid barrier (monotonic two-level counters, agent scope), then EVERY block adds the partial sums
// in block order, so every block subtracts the same h_j, bit for bit, and the result is reproducible run to run.
// Same arithmetic as orthog.hpp:41-62 + iterative.cpp:629-633 (h_j = (w, v_j) of the UPDATED w, w -= h_j v_j, then the norm).
// Launched as a cooperative kernel (every block resident: the barrier cannot dead-lock); one rank only (the sum over ranks of the
// chained form is a stream operation between two launches).
constexpr int kResBlk = 512, kResMaxGrid = 1024;
struct ResArgs {
  double *wr, *wi;
  const double *const *vr, *const *vi;
  int m, normalize;
  long long n;
  int prefetch;     // 1: v_{j+1} travels to LDS while the grid barrier of step j is pending (PALACE_AMD_GS_PREFETCH=0: off)
  double *partial;  // [2 parities][2 components][kResMaxGrid]
  unsigned *bar;    // group counters at [16 g], g < 16 (one 64-byte line each); the top counter at [256]; zero at launch
  double *coef, *nrm2;
};

// sums v[0 .. NV) over the block; the result is valid in thread 0 only
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*sm)[kResBlk / 64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const double s = wave_sum(v[k]);
    if (lane == 0) sm[k][wave] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < kResBlk / 64; q++) t += sm[k][q];
      v[k] = t;
    }
  }
  __syncthreads();
}

// a 16-byte lane at `base + off` with `base` wave-uniform (scalar registers) and a 32-bit per-lane byte offset: one address register
// per thread for every slot of every vector, instead of a 64-bit pair per slot.  The basis pointers are read from memory: the cast
// tells the compiler they are device memory (global_load, not flat_load).
#define PA_GLOBAL __attribute__((address_space(1)))
typedef double d2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long scalar_u64(unsigned long long u) {  // (pins a wave-uniform value to scalar registers)
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double2 ld_lane(const double *base, unsigned off) {
  const d2n v = *reinterpret_cast<const PA_GLOBAL d2n *>((const PA_GLOBAL char *)scalar_u64((unsigned long long)base) + off);
  return double2{v.x, v.y};
}
__device__ __forceinline__ void st_lane(double *base, unsigned off, double2 v) {
  *reinterpret_cast<PA_GLOBAL d2n *>((PA_GLOBAL char *)scalar_u64((unsigned long long)base) + off) = d2n{v.x, v.y};
}

// slots of v_{j+1} that wait in LDS (16-byte lanes per thread and part): 144 KB of the 160 KB of a compute unit at the largest
template <bool CPLX, int R>
constexpr int res_lds_slots() { return R < (CPLX ? 9 : 18) ? R : (CPLX ? 9 : 18); }
template <bool CPLX, int R>
constexpr size_t res_lds_bytes() { return (size_t)res_lds_slots<CPLX, R>() * (CPLX ? 2 : 1) * kResBlk * 16; }

template <bool CPLX, int R>
__global__ __launch_bounds__(kResBlk) void k_mgs_resident(const ResArgs A) {
  constexpr int NC = CPLX ? 2 : 1;
  constexpr int LS = res_lds_slots<CPLX, R>();
  extern __shared__ __attribute__((aligned(16))) double2 pre[];  // [LS][NC][kResBlk]: every thread reads what it wrote itself
  __shared__ double sm[2][kResBlk / 64];
  __shared__ double hb[2];
  const bool pf = A.prefetch != 0;
  const long long nv = A.n / 2, stride = (long long)gridDim.x * kResBlk;
  const long long i0 = (long long)blockIdx.x * kResBlk + threadIdx.x;
  const bool tail = (A.n & 1) && blockIdx.x == 0 && threadIdx.x == 0;  // the odd last entry
  const unsigned G = gridDim.x, grp = blockIdx.x & 15u, ng = (G - grp + 15u) / 16u, ngroups = G < 16u ? G : 16u;
  // slots r < R - 1 are in range for every thread (the host checks (R - 1) stride <= nv): only the last one is predicated -- a
  // select per slot keeps the loaded and the selected value alive side by side, 8 R registers more
  const bool in_last = i0 + (R - 1) * stride < nv;
  const unsigned o = 16u * (unsigned)i0;  // (bytes; i0 < 2^19: the per-slot offset r * stride goes into the scalar base)
  double2 a[R], b[R];
  double at = 0.0, bt = 0.0;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const bool in = r < R - 1 || in_last;
    a[r] = in ? ld_lane(A.wr + 2 * r * stride, o) : double2{0.0, 0.0};
    b[r] = (CPLX && in) ? ld_lane(A.wi + 2 * r * stride, o) : double2{0.0, 0.0};
  }
  if (tail) at = A.wr[A.n - 1], bt = CPLX ? A.wi[A.n - 1] : 0.0;
  // grid-wide sum of (v[0], v[1]) in step `step` (the barrier's round): the result in every thread of every block, the same bits
  // ... in two halves, so that the block can do something useful between its arrival at the barrier and the barrier's completion
  auto arrive = [&](double (&v)[2], const int step, const bool two) {
    block_sum<2>(v, sm);
    double *part = A.partial + (size_t)(step & 1) * 2 * kResMaxGrid;
    if (threadIdx.x == 0) {
      // (gfx9: one in-order counter covers loads and stores, agent-scope stores are written through -- grid_reduce above)
      __hip_atomic_store(&part[blockIdx.x], v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (two) __hip_atomic_store(&part[kResMaxGrid + blockIdx.x], v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // grid barrier: the last block of a group bumps the top counter; everybody waits for the top counter
      if (__hip_atomic_fetch_add(&A.bar[16u * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step * ng + ng - 1u)
        __hip_atomic_fetch_add(&A.bar[256], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto complete = [&](double (&v)[2], const int step, const bool two) {
    double *part = A.partial + (size_t)(step & 1) * 2 * kResMaxGrid;
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(&A.bar[256], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(step + 1) * ngroups)
        __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    // every block adds the partial sums of all blocks in the same order
    double h[2] = {0.0, 0.0};
    for (unsigned q = threadIdx.x; q < G; q += kResBlk) {
      h[0] += __hip_atomic_load(&part[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (two) h[1] += __hip_atomic_load(&part[kResMaxGrid + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    block_sum<2>(h, sm);
    if (threadIdx.x == 0) hb[0] = h[0], hb[1] = h[1];
    __syncthreads();
    v[0] = hb[0], v[1] = hb[1];
  };
  // the first LS slots of basis vector jj into this thread's LDS words, three slots at a time through spare registers
  auto prefetch = [&](const int jj) {
    const double *qr = A.vr[jj], *qi = CPLX ? A.vi[jj] : nullptr;
    constexpr int CH = 3;
#pragma unroll
    for (int r0 = 0; r0 < LS; r0 += CH) {
      double2 tr[CH], ti[CH];
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const int r = r0 + q;
        if (r < LS) {
          const bool in = r < R - 1 || in_last;
          tr[q] = in ? ld_lane(qr + 2 * r * stride, o) : double2{0.0, 0.0};
          ti[q] = (CPLX && in) ? ld_lane(qi + 2 * r * stride, o) : double2{0.0, 0.0};
        }
      }
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const int r = r0 + q;
        if (r < LS) {
          pre[(r * NC) * kResBlk + threadIdx.x] = tr[q];
          if (CPLX) pre[(r * NC + 1) * kResBlk + threadIdx.x] = ti[q];
        }
      }
    }
  };
  if (pf && A.m > 0) prefetch(0);
  for (int j = 0; j < A.m; j++) {
    const double *pr = A.vr[j], *pi = CPLX ? A.vi[j] : nullptr;
    double2 c[R], d[R];
    double ct = 0.0, dt = 0.0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      if (pf && r < LS) {  // (requested while the previous step's barrier was pending)
        c[r] = pre[(r * NC) * kResBlk + threadIdx.x];
        d[r] = CPLX ? pre[(r * NC + 1) * kResBlk + threadIdx.x] : double2{0.0, 0.0};
      } else {
        const bool in = r < R - 1 || in_last;
        c[r] = in ? ld_lane(pr + 2 * r * stride, o) : double2{0.0, 0.0};
        d[r] = (CPLX && in) ? ld_lane(pi + 2 * r * stride, o) : double2{0.0, 0.0};
      }
    }
    if (tail) ct = pr[A.n - 1], dt = CPLX ? pi[A.n - 1] : 0.0;
    double2 t0{0.0, 0.0}, t1{0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {  // (w, v_j) = v_j^H w (vector.cpp:674-685)
      t0 = mul_acc(t0, a[r], c[r]);
      if (CPLX) {
        t0 = mul_acc(t0, b[r], d[r]);
        t1 = mul_sub(mul_acc(t1, b[r], c[r]), a[r], d[r]);
      }
    }
    double v[2] = {hsum(t0), hsum(t1)};
    if (tail) {
      v[0] += at * ct;
      if (CPLX) v[0] += bt * dt, v[1] += bt * ct - at * dt;
    }
    arrive(v, j, CPLX);
    if (pf && j + 1 < A.m) prefetch(j + 1);  // (v_j is in registers: the LDS words are free)
    complete(v, j, CPLX);
    const double hr = v[0], hi = v[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      A.coef[NC * j] = hr;
      if (CPLX) A.coef[NC * j + 1] = hi;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {  // (hr + i hi)(c + i d) = (hr c - hi d) + i (hr d + hi c)
      a[r] = mul_sub(a[r], double2{hr, hr}, c[r]);
      if (CPLX) {
        a[r] = mul_acc(a[r], double2{hi, hi}, d[r]);
        b[r] = mul_sub(mul_sub(b[r], double2{hr, hr}, d[r]), double2{hi, hi}, c[r]);
      }
    }
    if (tail) {
      at -= hr * ct;
      if (CPLX) at += hi * dt, bt = bt - hr * dt - hi * ct;
    }
  }
  double s = 1.0;
  if (A.normalize) {  // iterative.cpp:632-633: w *= 1 / ||w||
    double2 t0{0.0, 0.0};
#pragma unroll
    for (int r = 0; r < R; r++) {
      t0 = mul_acc(t0, a[r], a[r]);
      if (CPLX) t0 = mul_acc(t0, b[r], b[r]);
    }
    double v[2] = {hsum(t0), 0.0};
    if (tail) {
      v[0] += at * at;
      if (CPLX) v[0] += bt * bt;
    }
    arrive(v, A.m, false);
    complete(v, A.m, false);
    if (blockIdx.x == 0 && threadIdx.x == 0) A.nrm2[0] = v[0];
    s = 1.0 / sqrt(fabs(v[0]));
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    if (r < R - 1 || in_last) {
      st_lane(A.wr + 2 * r * stride, o, double2{a[r].x * s, a[r].y * s});
      if (CPLX) st_lane(A.wi + 2 * r * stride, o, double2{b[r].x * s, b[r].y * s});
    }
  }
  if (tail) {
    A.wr[A.n - 1] = at * s;
    if (CPLX) A.wi[A.n - 1] = bt * s;
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

// device scratch: [counter | partial sums 2 kGB kMaxBlk | barrier words of the resident form (160 doubles) | coefficients pass 1
// (NC m) | pass 2 (NC m) | ||w||^2 | pad | basis pointers of the resident form (2 m)]; pinned: [the H column as on the device |
// basis pointers]
constexpr size_t kBarDoubles = 160;
struct GsBuffers {
  unsigned *counter;
  double *partial, *coef1, *coef2, *nrm2, *host;
  unsigned *bar;
  const double **ptrs, **host_ptrs;
};
GsBuffers buffers(const Context &c, int m) {
  Workspace &w = c.Work();
  static_assert(sizeof(double *) == sizeof(double), "pointer lists share the double scratch");
  const size_t head = 2 + (size_t)2 * kGB * kMaxBlk + kBarDoubles;
  double *d = w.GsDevice(head + 6 * (size_t)m + 2);
  double *h = w.GsPinned(6 * (size_t)m + 2);  // (m >= 1 here)
  return {reinterpret_cast<unsigned *>(d), d + 2, d + head, d + head + 2 * (size_t)m, d + head + 4 * (size_t)m, h,
          reinterpret_cast<unsigned *>(d + head - kBarDoubles), reinterpret_cast<const double **>(d + head + 4 * (size_t)m + 2),
          reinterpret_cast<const double **>(h + 4 * (size_t)m + 2)};
}

// The resident form (k_mgs_resident): true if it ran.  One block of 1 024 threads per compute unit, R 16-byte lanes of w (and of
// v_j) per thread; the smallest R that holds the vector is taken, a vector that does not fit keeps the chained form.
long long g_resident_columns = 0;
bool g_resident_failed = false;  // the cooperative launch has failed once: the chained form from then on
bool resident_mode() {           // PALACE_AMD_GS_RESIDENT=0 (read at every column: A / B runs in one process) keeps the chained form
  const char *e = std::getenv("PALACE_AMD_GS_RESIDENT");
  return !g_resident_failed && !(e && e[0] == '0');
}
template <bool CPLX, int R>
int resident_capacity_blocks() {  // blocks of k_mgs_resident<CPLX, R> the device holds at once (0: no cooperative launch)
  static int cap = -1;
  if (cap < 0) {
    int dev = 0, coop = 0, nb = 0, n_cu = 0;
    PA_HIP(hipGetDevice(&dev));
    PA_HIP(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    PA_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    // (the LDS words of the prefetch are dynamic: above 64 KB they have to be asked for)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_mgs_resident<CPLX, R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)res_lds_bytes<CPLX, R>()) != hipSuccess) {
      (void)hipGetLastError();
      coop = 0;
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mgs_resident<CPLX, R>, kResBlk, res_lds_bytes<CPLX, R>()) != hipSuccess) nb = 0;
    cap = coop ? std::min(n_cu * nb, kResMaxGrid) : 0;
  }
  return cap;
}
template <bool CPLX, int R>
bool launch_resident(const Context &c, const ResArgs &A0, long long nv, bool &fits) {
  const long long need = std::max<long long>(1, (nv + (long long)kResBlk * R - 1) / ((long long)kResBlk * R));
  fits = need <= resident_capacity_blocks<CPLX, R>();
  if (!fits) return false;
  if ((long long)(R - 1) * need * kResBlk > nv) return false;  // (the kernel predicates its last slot only; cannot happen for the
                                                                // smallest R that fits unless the device holds very few blocks)
  ResArgs A = A0;
  PA_HIP(hipMemsetAsync(A.bar, 0, 257 * sizeof(unsigned), c.stream));
  void *args[] = {&A};
  const hipError_t rc = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&k_mgs_resident<CPLX, R>), dim3((unsigned)need),
                                                   dim3(kResBlk), args, (unsigned)res_lds_bytes<CPLX, R>(), c.stream);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    g_resident_failed = true;
    return false;
  }
  return true;
}
template <bool CPLX>
bool run_resident(const Context &c, const Column &col, bool normalize, const GsBuffers &B) {
  const int m = col.m;
  for (int j = 0; j < m; j++) B.host_ptrs[j] = col.vr[j], B.host_ptrs[m + j] = CPLX ? col.vi[j] : nullptr;
  PA_HIP(hipMemcpyAsync(B.ptrs, B.host_ptrs, sizeof(double *) * 2 * (size_t)m, hipMemcpyHostToDevice, c.stream));
  const char *pe = std::getenv("PALACE_AMD_GS_PREFETCH");
  const ResArgs A{col.wr, col.wi, B.ptrs, B.ptrs + m, m, normalize ? 1 : 0, col.n, (pe && pe[0] == '0') ? 0 : 1, B.partial, B.bar, B.coef1, B.nrm2};
  const long long nv = col.n / 2;
  bool fits = false;
#define PA_TRY_RESIDENT(R)                                       \
  {                                                              \
    const bool ran = launch_resident<CPLX, R>(c, A, nv, fits);   \
    if (fits) return ran;                                        \
  }
  // the smallest register footprint that holds the vector (more blocks per compute unit while it is small)
  if constexpr (CPLX) {
    PA_TRY_RESIDENT(1) PA_TRY_RESIDENT(2) PA_TRY_RESIDENT(4) PA_TRY_RESIDENT(8) PA_TRY_RESIDENT(10) PA_TRY_RESIDENT(12)
  } else {
    PA_TRY_RESIDENT(1) PA_TRY_RESIDENT(2) PA_TRY_RESIDENT(4) PA_TRY_RESIDENT(8) PA_TRY_RESIDENT(16) PA_TRY_RESIDENT(20) PA_TRY_RESIDENT(24)
  }
#undef PA_TRY_RESIDENT
  return false;
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

  bool two_passes = false, resident = false;
  if (m == 0) {
    if (normalize) update(0, 0, B.coef1, TAIL_NORM, nullptr, nullptr, B.nrm2);
  } else if (kind == Orthogonalization::MGS && !weighted && !c.comm && wide && resident_mode() &&
             run_resident<CPLX>(c, col, normalize, B)) {
    resident = true;  // (w stayed in registers for the whole column, normalised there)
    g_resident_columns++;
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
  if (normalize && !resident) {
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
long long ResidentColumns() { return g_resident_columns; }
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
