// Device vector kernels and the Krylov / smoother / multigrid loop (see linalg.hpp for the
// reference symbols each class follows).  All kernels are 1-D streaming FP64 kernels, grid-stride,
// 16 B per lane; reductions use wavefront shuffles + one LDS stage and a second, single-block pass
// so the summation order (and with it iteration counts) is reproducible run to run.
#include "linalg.hpp"

#include <atomic>
#include <dlfcn.h>
#include <cmath>
#include <cstdlib>
#include <complex>
#include <cstdio>

#include "comm.hpp"
#include "krylov_impl.hpp"

namespace palace {

// ---- Vector -----------------------------------------------------------------------------------
Vector &Vector::operator=(Vector &&o) noexcept {
  if (this != &o) {
    if (own_ && d_) (void)hipFree(d_);
    d_ = o.d_, n_ = o.n_, own_ = o.own_;
    o.d_ = nullptr, o.n_ = 0, o.own_ = false;
  }
  return *this;
}
Vector::~Vector() {
  if (own_ && d_) (void)hipFree(d_);
}
void Vector::SetSize(int n) {
  if (own_ && n == n_) return;
  if (own_ && d_) (void)hipFree(d_);
  d_ = pa::dev_alloc<double>((size_t)n);
  n_ = n, own_ = true;
}
void Vector::MakeRef(double *ext, int n) {
  if (own_ && d_) (void)hipFree(d_);
  d_ = ext, n_ = n, own_ = false;
}

// ---- Workspace / StreamGraph -----------------------------------------------------------------
Workspace::~Workspace() {
  if (d_) (void)hipFree(d_);
  if (h_) (void)hipHostFree(h_);
  if (gs_d_) (void)hipFree(gs_d_);
  if (gs_h_) (void)hipHostFree(gs_h_);
  if (halo_stream_) (void)hipStreamDestroy(halo_stream_);
  if (ev_ready_) (void)hipEventDestroy(ev_ready_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
}
hipStream_t Workspace::HaloStream() {
  if (!halo_stream_) PA_HIP(hipStreamCreateWithFlags(&halo_stream_, hipStreamNonBlocking));
  return halo_stream_;
}
hipEvent_t Workspace::ReadyEvent() {
  if (!ev_ready_) PA_HIP(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
  return ev_ready_;
}
hipEvent_t Workspace::DoneEvent() {
  if (!ev_done_) PA_HIP(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
  return ev_done_;
}
double *Workspace::Device(size_t n) {
  PA_REQUIRE(n <= kDeviceDoubles, "reduction scratch request exceeds the workspace");
  if (!d_) d_ = pa::dev_alloc<double>(kDeviceDoubles);
  return d_;
}
double *Workspace::Pinned(size_t n) {
  PA_REQUIRE(n <= kPinnedDoubles, "pinned scratch request exceeds the workspace");
  if (!h_) PA_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_), kPinnedDoubles * sizeof(double), hipHostMallocDefault));
  return h_;
}

double *Workspace::GsDevice(size_t n) {
  if (n > gs_dn_) {  // (the stream that used the old buffer has been synchronised: every column ends with a copy to the host)
    if (gs_d_) {
      PA_HIP(hipDeviceSynchronize());
      (void)hipFree(gs_d_);
    }
    gs_dn_ = std::max<size_t>(2 * n, 32768);
    gs_d_ = pa::dev_alloc<double>(gs_dn_);
    PA_HIP(hipMemset(gs_d_, 0, gs_dn_ * sizeof(double)));
  }
  return gs_d_;
}
double *Workspace::GsPinned(size_t n) {
  if (n > gs_hn_) {
    if (gs_h_) {
      PA_HIP(hipDeviceSynchronize());
      (void)hipHostFree(gs_h_);
    }
    gs_hn_ = std::max<size_t>(2 * n, 1024);
    PA_HIP(hipHostMalloc(reinterpret_cast<void **>(&gs_h_), gs_hn_ * sizeof(double), hipHostMallocDefault));
  }
  return gs_h_;
}

StreamGraph::StreamGraph() {
  const char *e = std::getenv("PALACE_AMD_GRAPH");
  disabled_ = e && e[0] == '0';
}
StreamGraph::~StreamGraph() { Reset(); }
void StreamGraph::Reset() {
  if (exec_) (void)hipGraphExecDestroy(exec_);
  exec_ = nullptr, seen_ = 0, key_.clear();
}
static thread_local int tls_recording = 0;
bool StreamGraph::Recording() { return tls_recording > 0; }
void StreamGraph::RequireNotRecording(const char *what) {
  if (tls_recording > 0)
    throw pa::Error(std::string(what) + " waits for the device and cannot be part of a recorded launch sequence");
}
static bool graph_debug() {
  static const bool v = std::getenv("PALACE_AMD_GRAPH_DEBUG") != nullptr;
  return v;
}
bool StreamGraph::Capture(const Context &c, const std::function<void()> &body) {
  hipStream_t cap = c.stream;
  hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    if (graph_debug()) std::fprintf(stderr, "palace_amd graph: begin capture failed: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return false;
  }
  hipGraph_t g = nullptr;
  tls_recording++;
  try {
    body();
  } catch (const std::exception &ex) {
    // e.g. a solver inside the sequence that needs the host (RequireNotRecording): nothing has run, the sequence is
    // dropped and the caller runs it directly (where a genuine error shows up again)
    tls_recording--;
    if (graph_debug()) std::fprintf(stderr, "palace_amd graph: not recordable: %s\n", ex.what());
    (void)hipStreamEndCapture(cap, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return false;
  }
  tls_recording--;
  e = hipStreamEndCapture(cap, &g);
  if (e != hipSuccess || !g) {
    if (graph_debug()) std::fprintf(stderr, "palace_amd graph: end capture failed: %s\n", hipGetErrorString(e));
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return false;
  }
  e = hipGraphInstantiate(&exec_, g, nullptr, nullptr, 0);
  if (graph_debug()) {
    size_t nn = 0;
    (void)hipGraphGetNodes(g, nullptr, &nn);
    std::fprintf(stderr, "palace_amd graph: recorded %zu nodes, instantiate: %s\n", nn, hipGetErrorString(e));
  }
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    exec_ = nullptr;
    (void)hipGetLastError();
    return false;
  }
  return true;
}
namespace {
std::atomic<unsigned long long> g_config_epoch{1};

struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = std::getenv("PALACE_AMD_ROCTX");
    if (e && e[0] == '0') return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr, pop = nullptr;
      }
    }
  }
};
const Roctx &roctx() {
  static const Roctx r;
  return r;
}
}  // namespace
PhaseRange::PhaseRange(const char *name) : on_(roctx().push != nullptr) {
  if (on_) roctx().push(name);
}
PhaseRange::~PhaseRange() {
  if (on_) roctx().pop();
}
void StreamGraph::Invalidate() { g_config_epoch.fetch_add(1, std::memory_order_relaxed); }
unsigned long long StreamGraph::Epoch() { return g_config_epoch.load(std::memory_order_relaxed); }

void StreamGraph::Run(const Context &c, const std::vector<const void *> &key, const std::function<void()> &body) {
  if (!c.stream) return body();  // the null stream cannot be recorded
  // RCCL calls stay outside recordings (untested inside a capture here); the peer transport is plain kernels
  if (c.comm && !c.comm->GraphSafe()) return body();
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(c.stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return body();  // part of an enclosing recording (the V-cycle inside a PCG iteration)
  if (!disabled_) {
    if (key != key_ || epoch_ != Epoch()) Reset(), key_ = key, epoch_ = Epoch();
    if (!exec_ && seen_++ >= 1) {
      if (!Capture(c, body)) disabled_ = true;  // nothing of the recording ran: fall through to the direct form
    }
    if (exec_) {
      PA_HIP(hipGraphLaunch(exec_, c.stream));
      return;
    }
  }
  body();
}

// ---- kernels ----------------------------------------------------------------------------------
namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8

inline int grid_for(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  return (int)std::max(1LL, std::min<long long>(b, kMaxBlocks));
}
// Element-wise kernels: one lane entry per thread (no grid-stride round trips).  Measured on y = a x + b y over 2 x 512 MB
// (scripts/probes/stream_probe.hip): 5.87 TB/s against 5.11 TB/s for 2048 grid-striding blocks; reductions keep the
// bounded grid (their partial sums are one per block).
inline int grid_full(long long n) {
  long long b = (n + kBlock - 1) / kBlock;
  return (int)std::max(1LL, std::min<long long>(b, 1LL << 24));
}

#define PA_STRIDE_LOOP(i, n)                                                                  \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n);               \
       i += (long long)gridDim.x * blockDim.x)

// Streaming vector kernels move 16 bytes per lane (one dwordx4 access; 8-byte lanes reach ~0.6-0.7x of that
// rate on this memory system, MI355X_MICROARCH.md) whenever every pointer is 16-byte aligned; the scalar
// instantiation serves unaligned sub-vectors and the odd tail entry.
typedef double d2 __attribute__((ext_vector_type(2)));
template <int W>
struct Lane {
  using type = double;
};
template <>
struct Lane<2> {
  using type = d2;
};
template <int W, class Op>
__global__ __launch_bounds__(256) void k_ew(const Op op, long long n) {
  using T = typename Lane<W>::type;
  const long long nv = n / W;
  PA_STRIDE_LOOP(i, nv) op.template at<T>(i);
  if (W > 1) {
    const long long j = nv * W + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) op.template at<double>(j);
  }
}
template <class T>
__device__ __forceinline__ T *as(double *p) {
  return reinterpret_cast<T *>(p);
}
template <class T>
__device__ __forceinline__ const T *as(const double *p) {
  return reinterpret_cast<const T *>(p);
}
inline uintptr_t bits(const void *p) { return reinterpret_cast<uintptr_t>(p); }

struct OpFill {
  double *x;
  double s;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(x)[i] = T{} + s;
  }
  uintptr_t align() const { return bits(x); }
};
struct OpAxpy {
  double a;
  const double *x;
  double *y;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(y)[i] += a * as<T>(x)[i];
  }
  uintptr_t align() const { return bits(x) | bits(y); }
};
struct OpAxpby {
  double a;
  const double *x;
  double b;
  double *y;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(y)[i] = a * as<T>(x)[i] + b * as<T>(y)[i];
  }
  uintptr_t align() const { return bits(x) | bits(y); }
};
struct OpAxpbypcz {
  double a;
  const double *x;
  double b;
  const double *y;
  double g;
  double *z;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(z)[i] = a * as<T>(x)[i] + b * as<T>(y)[i] + g * as<T>(z)[i];
  }
  uintptr_t align() const { return bits(x) | bits(y) | bits(z); }
};
struct OpDiagMult {  // y = a d .* x (+ y)
  double a;
  const double *d, *x;
  double *y;
  int add;
  template <class T>
  __device__ void at(long long i) const {
    const T v = a * (as<T>(d)[i] * as<T>(x)[i]);
    as<T>(y)[i] = add ? as<T>(y)[i] + v : v;
  }
  uintptr_t align() const { return bits(d) | bits(x) | bits(y); }
};
struct OpScale {
  const double *d;
  double *y;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(y)[i] *= as<T>(d)[i];
  }
  uintptr_t align() const { return bits(d) | bits(y); }
};
struct OpRecip {
  double *x;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(x)[i] = 1.0 / as<T>(x)[i];
  }
  uintptr_t align() const { return bits(x); }
};
struct OpScal {
  double a;
  double *x;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(x)[i] *= a;
  }
  uintptr_t align() const { return bits(x); }
};
struct OpCheb0 {
  double sr;
  const double *di, *r;
  double *d;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(d)[i] = sr * as<T>(di)[i] * as<T>(r)[i];
  }
  uintptr_t align() const { return bits(di) | bits(r) | bits(d); }
};
struct OpChebK {
  double sd, sr;
  const double *di, *r;
  double *d;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(d)[i] = sd * as<T>(d)[i] + sr * as<T>(di)[i] * as<T>(r)[i];
  }
  uintptr_t align() const { return bits(di) | bits(r) | bits(d); }
};
// one Chebyshev step around the operator apply t = A d (chebyshev.cpp:208-216 fused into one pass):
//   y += d;  r -= t;  d = sd d + sr dinv .* r
struct OpChebStep {
  double sd, sr;
  const double *di, *t;
  double *r, *d, *y;
  template <class T>
  __device__ void at(long long i) const {
    const T dv = as<T>(d)[i], yv = as<T>(y)[i], dinv = as<T>(di)[i];  // every load before the first store
    const T rv = as<T>(r)[i] - as<T>(t)[i];
    as<T>(y)[i] = yv + dv;
    as<T>(r)[i] = rv;
    as<T>(d)[i] = sd * dv + sr * dinv * rv;
  }
  uintptr_t align() const { return bits(di) | bits(t) | bits(r) | bits(d) | bits(y); }
};
// the same step written for the accumulated correction e_k = d_0 + ... + d_{k-1} (so d_{k-1} = e_k - e_{k-1} and r_k = r_0 - A e_k):
//   out (+)= e_k + sd (e_k - e_{k-1}) + sr dinv .* (r_0 - t),  t = A e_k
// r_0 is only read (it can be the caller's right-hand side itself), y is touched once, at the last step: 48 instead of 64 bytes
// per entry and step.  ep == nullptr: e_{k-1} = 0 (the first step).  `out` may be the buffer of e_{k-1}.
struct OpChebStep3 {
  double sd, sr;
  const double *di, *t, *r0, *ek, *ep;
  double *out;
  int add;
  template <class T>
  __device__ void at(long long i) const {
    const T e = as<T>(ek)[i], dinv = as<T>(di)[i], rv = as<T>(r0)[i] - as<T>(t)[i];
    T dk = sr * dinv * rv;
    if (ep) dk += sd * (e - as<T>(ep)[i]);
    else dk += sd * e;
    T v = e + dk;
    if (add) v += as<T>(out)[i];
    as<T>(out)[i] = v;
  }
  uintptr_t align() const { return bits(di) | bits(t) | bits(r0) | bits(ek) | bits(ep) | bits(out); }
};
__device__ __forceinline__ double vsqrt(double v) { return sqrt(v); }
__device__ __forceinline__ d2 vsqrt(d2 v) { return d2{sqrt(v.x), sqrt(v.y)}; }
// x = sqrt(s x)   (linalg::Sqrt, vector.cpp:774-781)
struct OpSqrt {
  double s;
  double *x;
  template <class T>
  __device__ void at(long long i) const {
    as<T>(x)[i] = vsqrt(as<T>(x)[i] * s);
  }
  uintptr_t align() const { return bits(x); }
};
// x += a p;  r -= a z   (the two AXPYs of a CG iteration, iterative.cpp:448-449)
struct OpCgUpdate {
  double a;
  const double *p, *z;
  double *x, *r;
  template <class T>
  __device__ void at(long long i) const {
    const T xv = as<T>(x)[i] + a * as<T>(p)[i], rv = as<T>(r)[i] - a * as<T>(z)[i];
    as<T>(x)[i] = xv;
    as<T>(r)[i] = rv;
  }
  uintptr_t align() const { return bits(p) | bits(z) | bits(x) | bits(r); }
};
__global__ void k_set_sub(double *x, const int32_t *__restrict__ rows, int n, double s) {
  PA_STRIDE_LOOP(i, n) x[rows[i]] = s;
}
__global__ void k_set_sub_vec(double *x, const int32_t *__restrict__ rows, int n, const double *__restrict__ y) {
  PA_STRIDE_LOOP(i, n) x[rows[i]] = y[rows[i]];
}
__global__ void k_random(double *x, long long n, uint64_t seed) {
  PA_STRIDE_LOOP(i, n) {
    // splitmix64 on (seed, i): counter-based, reproducible for any launch shape
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)i + 0x632BE59BD9B4E019ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    x[i] = 2.0 * ((double)(z >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double part[kBlock / 64];
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) part[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < kBlock / 64; i++) t += part[i];
  return t;  // valid on thread 0
}

__device__ __forceinline__ double hsum(double v) { return v; }
__device__ __forceinline__ double hsum(d2 v) { return v.x + v.y; }

template <int W>
__global__ __launch_bounds__(256) void k_dot_partial(const double *__restrict__ x, const double *__restrict__ y,
                                                     long long n, double *__restrict__ partial) {
  using T = typename Lane<W>::type;
  const long long nv = n / W;
  T acc = T{};
  PA_STRIDE_LOOP(i, nv) acc += as<T>(x)[i] * as<T>(y)[i];
  double s = hsum(acc);
  if (W > 1) {
    const long long j = nv * W + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) s += x[j] * y[j];
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// sum of the entries (linalg::LocalSum, vector.cpp:687-699): same two-stage reduction as the dot product
template <int W>
__global__ __launch_bounds__(256) void k_sum_partial(const double *__restrict__ x, long long n,
                                                     double *__restrict__ partial) {
  using T = typename Lane<W>::type;
  const long long nv = n / W;
  T acc = T{};
  PA_STRIDE_LOOP(i, nv) acc += as<T>(x)[i];
  double s = hsum(acc);
  if (W > 1) {
    const long long j = nv * W + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) s += x[j];
  }
  s = block_sum(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void k_dot_final(const double *__restrict__ partial, int nb, double *__restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[i];
  s = block_sum(s);
  if (threadIdx.x == 0) out[0] = s;
}

// Batched form for classical Gram-Schmidt (orthog.hpp:57-89): up to kDotBatch inner products (w, V_j) in
// one pass over w, and w -= sum_j h_j V_j in one pass; one all-reduce for the whole column.
constexpr int kDotBatch = 8;
struct VecPtrs {
  const double *v[kDotBatch];
};
struct Coefs {
  double h[kDotBatch];
};
template <int W>
__global__ __launch_bounds__(256) void k_multi_dot_partial(const double *__restrict__ w, const VecPtrs V, int m,
                                                           long long n, double *__restrict__ partial) {
  using T = typename Lane<W>::type;
  const long long nv = n / W;
  T acc[kDotBatch];
#pragma unroll
  for (int j = 0; j < kDotBatch; j++) acc[j] = T{};
  PA_STRIDE_LOOP(i, nv) {
    const T wi = as<T>(w)[i];
#pragma unroll
    for (int j = 0; j < kDotBatch; j++)
      if (j < m) acc[j] += wi * as<T>(V.v[j])[i];
  }
  double s[kDotBatch];
#pragma unroll
  for (int j = 0; j < kDotBatch; j++) s[j] = hsum(acc[j]);
  if (W > 1) {
    const long long i = nv * W + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
#pragma unroll
      for (int j = 0; j < kDotBatch; j++)
        if (j < m) s[j] += w[i] * V.v[j][i];
    }
  }
#pragma unroll
  for (int j = 0; j < kDotBatch; j++) {
    const double t = block_sum(s[j]);
    if (threadIdx.x == 0) partial[(size_t)j * gridDim.x + blockIdx.x] = t;
    __syncthreads();
  }
}
__global__ void k_multi_dot_final(const double *__restrict__ partial, int nb, int m, double *__restrict__ out) {
  for (int j = 0; j < m; j++) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[(size_t)j * nb + i];
    s = block_sum(s);
    if (threadIdx.x == 0) out[j] = s;
    __syncthreads();
  }
}
struct OpMultiAxpy {
  Coefs a;
  VecPtrs V;
  int m;
  double *w;
  template <class T>
  __device__ void at(long long i) const {
    T wi = as<T>(w)[i];
#pragma unroll
    for (int j = 0; j < kDotBatch; j++)
      if (j < m) wi -= a.h[j] * as<T>(V.v[j])[i];
    as<T>(w)[i] = wi;
  }
  uintptr_t align() const {
    uintptr_t b = bits(w);
    for (int j = 0; j < m; j++) b |= bits(V.v[j]);
    return b;
  }
};

// ---- PCG with device-resident scalars (CgSolver::MultDevice) ---------------------------------------------
// st[]: the scalars of iterative.cpp:375-486 in device memory
enum CgSlot { CG_BETA = 0, CG_BETA_PREV, CG_DENOM, CG_RES, CG_EPS, CG_INIT, CG_STOP, CG_IT, CG_BAD, CG_TMP, CG_NSLOT = 16 };
enum CgStep { CG_STEP_RHS = 0, CG_STEP_START, CG_STEP_DENOM, CG_STEP_BETA };
struct CgTol {
  double rel, abs;
  int use_rhs;  // initial residual from the right-hand side (initial guess given): 1 = (B b, b), 2 = no preconditioner
};
template <int STEP>
__device__ __forceinline__ void cg_scalar_step(double *__restrict__ st, const double v, const CgTol tol) {
  if (STEP == CG_STEP_RHS) {
    // (B b, b) -> sqrt |.|; without a preconditioner the reference takes sqrt |Norml2(b)| (iterative.cpp:406-411: the
    // norm, not its square, goes under the root) and so does MultHost -- v = (b, b) here
    st[CG_INIT] = tol.use_rhs == 2 ? sqrt(sqrt(fabs(v))) : sqrt(fabs(v));
  } else if (STEP == CG_STEP_START) {  // beta = (z, r), iterative.cpp:400-421
    const double res = sqrt(fabs(v));
    st[CG_BETA] = v, st[CG_BETA_PREV] = v, st[CG_RES] = res, st[CG_IT] = 0.0;
    if (!tol.use_rhs) st[CG_INIT] = res;
    const double eps = fmax(tol.rel * st[CG_INIT], tol.abs);
    st[CG_EPS] = eps;
    const bool bad = !isfinite(v);
    st[CG_BAD] = bad ? 1.0 : 0.0;
    st[CG_STOP] = (bad || res < eps) ? 1.0 : 0.0;
  } else if (st[CG_STOP] == 0.0) {
    if (STEP == CG_STEP_DENOM) {  // (A p, p), iterative.cpp:444
      st[CG_DENOM] = v;
      if (!isfinite(v)) st[CG_BAD] = 2.0, st[CG_STOP] = 1.0;
    } else {  // beta = (z, r) of the new residual, iterative.cpp:460-474
      const double res = sqrt(fabs(v));
      st[CG_BETA_PREV] = st[CG_BETA], st[CG_BETA] = v, st[CG_RES] = res, st[CG_IT] += 1.0;
      if (!isfinite(v)) st[CG_BAD] = 1.0, st[CG_STOP] = 1.0;
      if (res < st[CG_EPS]) st[CG_STOP] = 1.0;
    }
  }
}
// second stage of the inner product fused with the scalar update (single process)
template <int STEP>
__global__ void k_cg_dot_final(const double *__restrict__ partial, int nb, double *__restrict__ st, const CgTol tol) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partial[i];
  s = block_sum(s);
  if (threadIdx.x == 0) cg_scalar_step<STEP>(st, s, tol);
}
// the same update after the all-reduce of st[CG_TMP] (multi-rank)
template <int STEP>
__global__ void k_cg_scalar(double *__restrict__ st, const CgTol tol) {
  cg_scalar_step<STEP>(st, st[CG_TMP], tol);
}
// element-wise kernels whose coefficient comes from st[]; they leave the vectors alone once the solve has stopped
template <int W, class Op>
__global__ __launch_bounds__(256) void k_ew_cg(Op op, long long n, const double *__restrict__ st) {
  if (st[CG_STOP] != 0.0) return;
  op.load(st);
  using T = typename Lane<W>::type;
  const long long nv = n / W;
  PA_STRIDE_LOOP(i, nv) op.template at<T>(i);
  if (W > 1) {
    const long long j = nv * W + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) op.template at<double>(j);
  }
}
struct OpCgDirDev : OpAxpby {  // p = z + (beta / beta_prev) p, iterative.cpp:436-441
  __device__ void load(const double *st) { a = 1.0, b = st[CG_BETA] / st[CG_BETA_PREV]; }
};
struct OpCgUpdateDev : OpCgUpdate {  // alpha = beta / (A p, p); x += alpha p; r -= alpha z, iterative.cpp:446-449
  __device__ void load(const double *st) { a = st[CG_BETA] / st[CG_DENOM]; }
};

// layout of the context's reduction scratch: [kDotBatch * kMaxBlocks] partial sums, then kMaxBlocks + kDotBatch results
struct Scratch {
  double *d_partial, *h_result;
};
Scratch scratch(const Context &c) {
  Workspace &w = c.Work();
  return {w.Device((size_t)kDotBatch * kMaxBlocks + kMaxBlocks + kDotBatch), w.Pinned(kDotBatch)};
}

#define PA_LAUNCH(kernel, n, stream, ...)                                                  \
  do {                                                                                     \
    if ((n) > 0) {                                                                         \
      hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, __VA_ARGS__); \
      PA_HIP(hipGetLastError());                                                           \
    }                                                                                      \
  } while (0)

// 16-byte lanes when every pointer allows it; the grid covers the lane count
template <class Op>
void launch_ew(const Op &op, long long n, hipStream_t stream) {
  if (n <= 0) return;
  if ((op.align() & 15) == 0 && n >= 2)
    hipLaunchKernelGGL((k_ew<2, Op>), dim3(grid_full((n + 1) / 2)), dim3(kBlock), 0, stream, op, n);
  else
    hipLaunchKernelGGL((k_ew<1, Op>), dim3(grid_full(n)), dim3(kBlock), 0, stream, op, n);
  PA_HIP(hipGetLastError());
}

}  // namespace

namespace linalg {

void Copy(const Context &c, const Vector &x, Vector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in Copy");
  if (x.Data() != y.Data())
    PA_HIP(hipMemcpyAsync(y.Data(), x.Data(), sizeof(double) * (size_t)x.Size(), hipMemcpyDeviceToDevice, c.stream));
}
void Fill(const Context &c, Vector &x, double s) {
  if (s == 0.0)
    PA_HIP(hipMemsetAsync(x.Data(), 0, sizeof(double) * (size_t)x.Size(), c.stream));
  else
    launch_ew(OpFill{x.Data(), s}, x.Size(), c.stream);
}
void AXPY(const Context &c, double a, const Vector &x, Vector &y) {
  launch_ew(OpAxpy{a, x.Data(), y.Data()}, x.Size(), c.stream);
}
void AXPBY(const Context &c, double a, const Vector &x, double b, Vector &y) {
  launch_ew(OpAxpby{a, x.Data(), b, y.Data()}, x.Size(), c.stream);
}
void AXPBYPCZ(const Context &c, double a, const Vector &x, double b, const Vector &y, double g, Vector &z) {
  launch_ew(OpAxpbypcz{a, x.Data(), b, y.Data(), g, z.Data()}, x.Size(), c.stream);
}
void SetSubVector(const Context &c, Vector &x, const int32_t *rows, int n, double s) {
  PA_LAUNCH(k_set_sub, n, c.stream, x.Data(), rows, n, s);
}
void SetSubVector(const Context &c, Vector &x, const int32_t *rows, int n, const Vector &y) {
  PA_LAUNCH(k_set_sub_vec, n, c.stream, x.Data(), rows, n, y.Data());
}
void Scale(const Context &c, const Vector &d, Vector &y) {
  launch_ew(OpScale{d.Data(), y.Data()}, y.Size(), c.stream);
}
void Reciprocal(const Context &c, Vector &x) { launch_ew(OpRecip{x.Data()}, x.Size(), c.stream); }
void Scale(const Context &c, double s, Vector &x) { launch_ew(OpScal{s, x.Data()}, x.Size(), c.stream); }

double Dot(const Context &c, const Vector &x, const Vector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in Dot");
  StreamGraph::RequireNotRecording("linalg::Dot");
  const Scratch s = scratch(c);
  const bool wide = ((bits(x.Data()) | bits(y.Data())) & 15) == 0 && x.Size() >= 2;
  const int nb = grid_for(wide ? (x.Size() + 1) / 2 : x.Size());
  if (wide)
    hipLaunchKernelGGL(k_dot_partial<2>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), y.Data(), (long long)x.Size(),
                       s.d_partial);
  else
    hipLaunchKernelGGL(k_dot_partial<1>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), y.Data(), (long long)x.Size(),
                       s.d_partial);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kBlock), 0, c.stream, s.d_partial, nb, s.d_partial + kMaxBlocks);
  PA_HIP(hipGetLastError());
  if (c.comm) c.comm->AllReduceSum(s.d_partial + kMaxBlocks, 1, c.stream);  // Mpi::GlobalSum
  PA_HIP(hipMemcpyAsync(s.h_result, s.d_partial + kMaxBlocks, sizeof(double), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
  if (c.comm) c.comm->PeerCheckNow();  // (a timed-out wait of the peer transport surfaces here, not as a wrong sum)
  return s.h_result[0];
}
// H[j] = (w, V[j]) for j < m (global), batches of kDotBatch; w -= sum_j H[j] V[j] if `subtract`
void MultiDot(const Context &c, const Vector &w, const std::vector<Vector> &V, int m, double *H) {
  StreamGraph::RequireNotRecording("linalg::MultiDot");
  const Scratch s = scratch(c);
  const int nb = grid_for(w.Size());
  double *d_out = s.d_partial + (size_t)kDotBatch * kMaxBlocks;
  for (int j0 = 0; j0 < m; j0 += kDotBatch) {
    const int mb = std::min(kDotBatch, m - j0);
    VecPtrs P{};
    for (int j = 0; j < mb; j++) P.v[j] = V[j0 + j].Data();
    uintptr_t al = bits(w.Data());
    for (int j = 0; j < mb; j++) al |= bits(P.v[j]);
    if ((al & 15) == 0)
      hipLaunchKernelGGL(k_multi_dot_partial<2>, dim3(nb), dim3(kBlock), 0, c.stream, w.Data(), P, mb,
                         (long long)w.Size(), s.d_partial);
    else
      hipLaunchKernelGGL(k_multi_dot_partial<1>, dim3(nb), dim3(kBlock), 0, c.stream, w.Data(), P, mb,
                         (long long)w.Size(), s.d_partial);
    hipLaunchKernelGGL(k_multi_dot_final, dim3(1), dim3(kBlock), 0, c.stream, s.d_partial, nb, mb, d_out);
    PA_HIP(hipGetLastError());
    if (c.comm) c.comm->AllReduceSum(d_out, mb, c.stream);
    PA_HIP(hipMemcpyAsync(s.h_result, d_out, sizeof(double) * mb, hipMemcpyDeviceToHost, c.stream));
    PA_HIP(hipStreamSynchronize(c.stream));
    if (c.comm) c.comm->PeerCheckNow();  // (a timed-out wait of the peer transport surfaces here, not as a wrong sum)
    for (int j = 0; j < mb; j++) H[j0 + j] = s.h_result[j];
  }
}
void MultiAXPY(const Context &c, const double *H, const std::vector<Vector> &V, int m, Vector &w) {
  for (int j0 = 0; j0 < m; j0 += kDotBatch) {
    const int mb = std::min(kDotBatch, m - j0);
    VecPtrs P{};
    Coefs a{};
    for (int j = 0; j < mb; j++) P.v[j] = V[j0 + j].Data(), a.h[j] = H[j0 + j];
    launch_ew(OpMultiAxpy{a, P, mb, w.Data()}, w.Size(), c.stream);
  }
}
void OrthogonalizeColumn(const Context &c, Orthogonalization kind, const std::vector<Vector> &V, Vector &w, double *H,
                         int m, const Operator *weight) {
  PA_REQUIRE(m >= 0 && (size_t)m <= V.size(), "Out of bounds number of columns for orthogonalization!");
  if (m == 0) return;
  for (int j = 0; j < m; j++) PA_REQUIRE(V[j].Size() == w.Size(), "size mismatch in OrthogonalizeColumn");
  Vector ws;
  if (weight) {
    PA_REQUIRE(weight->Height() == w.Size() && weight->Width() == w.Size(), "weight operator does not match the vectors");
    ws.SetSize(w.Size());
  }
  if (DeviceOrthogonalization() && !(weight && kind == Orthogonalization::MGS)) {  // orthog.hip: coefficients stay on the device
    const int passes = (weight && kind == Orthogonalization::CGS2) ? 2 : 1;  // (W w of the refinement pass is recomputed here)
    const Orthogonalization k1 = weight ? Orthogonalization::CGS : kind;
    std::vector<double> dH;
    for (int pass = 0; pass < passes; pass++) {
      if (weight) weight->Mult(w, ws);
      if (pass) dH.resize((size_t)m);
      OrthogonalizeColumnDevice(c, k1, V, w, weight ? &ws : nullptr, pass ? dH.data() : H, m, false, nullptr);
    }
    for (size_t j = 0; j < dH.size(); j++) H[j] += dH[j];
    return;
  }
  if (kind == Orthogonalization::MGS) {  // orthog.hpp:41-55
    for (int j = 0; j < m; j++) {
      if (weight) weight->Mult(w, ws);
      H[j] = Dot(c, weight ? ws : w, V[j]);
      AXPY(c, -H[j], V[j], w);
    }
    return;
  }
  // classical Gram-Schmidt: all inner products of a pass from the same w, one reduction (orthog.hpp:57-89)
  if (weight) weight->Mult(w, ws);
  MultiDot(c, weight ? ws : w, V, m, H);
  MultiAXPY(c, H, V, m, w);
  if (kind == Orthogonalization::CGS2) {
    std::vector<double> dH((size_t)m);
    if (weight) weight->Mult(w, ws);
    MultiDot(c, weight ? ws : w, V, m, dH.data());
    MultiAXPY(c, dH.data(), V, m, w);
    for (int j = 0; j < m; j++) H[j] += dH[j];
  }
}
double Sum(const Context &c, const Vector &x) {
  StreamGraph::RequireNotRecording("linalg::Sum");
  const Scratch s = scratch(c);
  if (x.Size() == 0 && !c.comm) return 0.0;
  const bool wide = (bits(x.Data()) & 15) == 0 && x.Size() >= 2;
  const int nb = grid_for(wide ? (x.Size() + 1) / 2 : std::max(x.Size(), 1));
  if (wide)
    hipLaunchKernelGGL(k_sum_partial<2>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), (long long)x.Size(), s.d_partial);
  else
    hipLaunchKernelGGL(k_sum_partial<1>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), (long long)x.Size(), s.d_partial);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kBlock), 0, c.stream, s.d_partial, nb, s.d_partial + kMaxBlocks);
  PA_HIP(hipGetLastError());
  if (c.comm) c.comm->AllReduceSum(s.d_partial + kMaxBlocks, 1, c.stream);  // Mpi::GlobalSum (vector.hpp)
  PA_HIP(hipMemcpyAsync(s.h_result, s.d_partial + kMaxBlocks, sizeof(double), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
  if (c.comm) c.comm->PeerCheckNow();  // (a timed-out wait of the peer transport surfaces here, not as a wrong sum)
  return s.h_result[0];
}
void Sqrt(const Context &c, Vector &x, double s) { launch_ew(OpSqrt{s, x.Data()}, x.Size(), c.stream); }
double Norml2(const Context &c, const Vector &x) { return std::sqrt(Dot(c, x, x)); }
double Normalize(const Context &c, Vector &x) {
  const double nrm = Norml2(c, x);
  PA_REQUIRE(nrm > 0.0, "zero vector norm in normalization");
  launch_ew(OpScal{1.0 / nrm, x.Data()}, x.Size(), c.stream);
  return nrm;
}
void SetRandom(const Context &c, Vector &x, uint64_t seed) {
  const uint64_t rank_seed = seed + (c.comm ? 7919ull * (uint64_t)c.comm->Rank() : 0ull);
  PA_LAUNCH(k_random, x.Size(), c.stream, x.Data(), (long long)x.Size(), rank_seed);
}
void ChebyOrder0(const Context &c, double sr, const Vector &dinv, const Vector &r, Vector &d) {
  launch_ew(OpCheb0{sr, dinv.Data(), r.Data(), d.Data()}, d.Size(), c.stream);
}
void ChebyStep(const Context &c, double sd, double sr, const Vector &dinv, const Vector &t, Vector &r, Vector &d,
               Vector &y) {
  launch_ew(OpChebStep{sd, sr, dinv.Data(), t.Data(), r.Data(), d.Data(), y.Data()}, d.Size(), c.stream);
}
void ChebyStep3(const Context &c, double sd, double sr, const Vector &dinv, const Vector &t, const Vector &r0, const Vector &ek,
                const Vector *ep, Vector &out, bool add) {
  launch_ew(OpChebStep3{sd, sr, dinv.Data(), t.Data(), r0.Data(), ek.Data(), ep ? ep->Data() : nullptr, out.Data(), add ? 1 : 0},
            out.Size(), c.stream);
}
void CgUpdate(const Context &c, double a, const Vector &p, const Vector &z, Vector &x, Vector &r) {
  launch_ew(OpCgUpdate{a, p.Data(), z.Data(), x.Data(), r.Data()}, x.Size(), c.stream);
}
void ChebyOrderK(const Context &c, double sd, double sr, const Vector &dinv, const Vector &r, Vector &d) {
  launch_ew(OpChebK{sd, sr, dinv.Data(), r.Data(), d.Data()}, d.Size(), c.stream);
}

// Power iteration on D^{-1} A (chebyshev.cpp:14-28 + linalg/operator.cpp:583-631, herm = true)
double SpectralNorm(const Context &c, const Operator &A, const Vector &dinv, double tol, int max_it, uint64_t seed) {
  Vector u(A.Height()), v(A.Height());
  SetRandom(c, u, seed);
  Normalize(c, u);
  double l = 0.0, l0 = 0.0;
  int it = 0;
  while (it < max_it) {
    A.Mult(u, v);
    Scale(c, dinv, v);
    Copy(c, v, u);
    l = Normalize(c, u);
    if (it > 0 && std::abs(l - l0) / l0 < tol) break;
    l0 = l;
    it++;
  }
  return l;
}

}  // namespace linalg

// ---- Operator defaults ------------------------------------------------------------------------
void Operator::MultTranspose(const Vector &, Vector &) const { throw pa::Error("MultTranspose not implemented"); }
void Operator::AddMult(const Vector &, Vector &, double) const { throw pa::Error("AddMult not implemented"); }
void Operator::AddMultTranspose(const Vector &, Vector &, double) const { throw pa::Error("AddMultTranspose not implemented"); }
void Operator::AssembleDiagonal(Vector &) const { throw pa::Error("AssembleDiagonal not implemented"); }
void Operator::MultChebyStep(const Vector &, const ChebyStepArgs &) const { throw pa::Error("MultChebyStep not implemented"); }
void Operator::MultResidual(const Vector &, const Vector &, Vector *, const Vector *, double, Vector *) const {
  throw pa::Error("MultResidual not implemented");
}
void Solver::Mult2(const Vector &, Vector &, Vector &) const { throw pa::Error("Mult2 not implemented"); }

namespace ceed {
Operator::Operator(const Context &ctx, pa_op *op, bool own)
    : palace::Operator(pa_op_height(op), pa_op_width(op)), op_(op), own_(own), ctx_(&ctx) {}
Operator::~Operator() {
  if (own_) pa_op_destroy(op_);
}
static void check(int rc) {
  if (rc) throw pa::Error(pa_last_error());
}
static pa_op *create_op(int h, int w) {
  pa_op *op = nullptr;
  check(pa_op_create(h, w, &op));
  return op;
}
Operator::Operator(const Context &ctx, int h, int w) : palace::Operator(h, w), op_(create_op(h, w)), own_(true), ctx_(&ctx) {}
void Operator::AddSubOperator(pa_geom *geom, const pa_restriction_desc &restr, const pa_basis_desc &basis, int qfunction,
                              const void *qf_ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops) {
  check(pa_op_add_sub(op_, geom, &restr, &basis, qfunction, qf_ctx, ctx_size, trial_ops, test_ops));
}
void Operator::AddSubOperator(pa_geom *geom, const pa_restriction_desc &restr, const pa_dense_basis_desc &basis, int qfunction,
                              const void *qf_ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops) {
  check(pa_op_add_sub_dense(op_, geom, &restr, &basis, qfunction, qf_ctx, ctx_size, trial_ops, test_ops));
}
void Operator::Finalize() {
  StreamGraph::Invalidate();
  check(pa_op_finalize(op_));
}
void Operator::DestroyAssemblyData() const { check(pa_op_destroy_assembly_data(op_)); }
std::size_t Operator::Size() const { return (std::size_t)pa_op_num_sub(op_); }
void Operator::SetDofMultiplicity(Vector &&mult) {
  PA_REQUIRE(mult.Size() == 0 || mult.Size() == height, "dof multiplicity: one entry per row of the operator");
  StreamGraph::Invalidate();
  dof_multiplicity_ = std::move(mult);
}
void Operator::Mult(const Vector &x, Vector &y) const {
  check(pa_op_mult(op_, x.Data(), y.Data(), ctx_->stream));
  if (dof_multiplicity_.Size() > 0) linalg::Scale(*ctx_, dof_multiplicity_, y);  // operator.cpp:186-189
}
void Operator::AddMult(const Vector &x, Vector &y, double a) const {
  PA_REQUIRE(a == 1.0, "ceed::Operator::AddMult only supports coefficient = 1.0!");  // operator.cpp:194
  if (dof_multiplicity_.Size() > 0) {  // operator.cpp:195-207
    if (temp_.Size() != height) temp_.SetSize(height);
    check(pa_op_mult(op_, x.Data(), temp_.Data(), ctx_->stream));
    linalg::Scale(*ctx_, dof_multiplicity_, temp_);
    linalg::AXPY(*ctx_, 1.0, temp_, y);
    return;
  }
  check(pa_op_apply_add(op_, x.Data(), y.Data(), ctx_->stream));
}
void Operator::MultTranspose(const Vector &x, Vector &y) const {
  if (dof_multiplicity_.Size() > 0) {  // operator.cpp:213-217 -> :224-235: y = A^T (d .* x)
    if (temp_.Size() != height) temp_.SetSize(height);
    linalg::Copy(*ctx_, x, temp_);
    linalg::Scale(*ctx_, dof_multiplicity_, temp_);
    check(pa_op_mult_transpose(op_, temp_.Data(), y.Data(), ctx_->stream));
    return;
  }
  check(pa_op_mult_transpose(op_, x.Data(), y.Data(), ctx_->stream));
}
void Operator::AddMultTranspose(const Vector &x, Vector &y, double a) const {
  PA_REQUIRE(a == 1.0, "ceed::Operator::AddMultTranspose only supports coefficient = 1.0!");  // operator.cpp:219
  if (dof_multiplicity_.Size() > 0) {
    if (temp_.Size() != height) temp_.SetSize(height);
    linalg::Copy(*ctx_, x, temp_);
    linalg::Scale(*ctx_, dof_multiplicity_, temp_);
    check(pa_op_apply_add_transpose(op_, temp_.Data(), y.Data(), ctx_->stream));
    return;
  }
  check(pa_op_apply_add_transpose(op_, x.Data(), y.Data(), ctx_->stream));
}
bool Operator::IsSymmetric() const { return pa_op_is_symmetric(op_) != 0; }
void Operator::AssembleDiagonal(Vector &diag) const { check(pa_op_assemble_diagonal(op_, diag.Data(), ctx_->stream)); }
void Operator::MultComplex(const Operator &Ar, const Operator &Ai, const Vector &xr, const Vector &xi, Vector &yr, Vector &yi,
                           int ess_policy) {
  if (pa_op_mult_complex(Ar.op_, Ai.op_, xr.Data(), xi.Data(), yr.Data(), yi.Data(), ess_policy, Ar.ctx_->stream))
    throw pa::Error(pa_last_error());
}
void Operator::SetInterfaceDofs(const std::vector<int32_t> &ldofs) {
  StreamGraph::Invalidate();
  if (pa_op_set_interface_dofs(op_, ldofs.data(), (int32_t)ldofs.size())) throw pa::Error(pa_last_error());
}
void Operator::MultAfter(const Vector &x, Vector &y, hipEvent_t after) const {
  if (pa_op_mult_after(op_, x.Data(), y.Data(), ctx_->stream, after)) throw pa::Error(pa_last_error());
}
void Operator::SetEssential(const int32_t *ess_host, int n) {
  StreamGraph::Invalidate();
  check(pa_op_set_essential(op_, ess_host, n));
}
bool Operator::SupportsSplit() const { return pa_op_supports_split(op_) != 0; }
void Operator::MultSplit(const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *y, double *yg,
                         int n_true, int ess_policy) const {
  check(pa_op_mult_split(op_, x, xg0, xg1, sel, y, yg, n_true, ess_policy, ctx_->stream));
}
bool Operator::MultEssentialDiag(const Vector &x, Vector &y, bool diag_one) const {
  int handled = 0;
  check(pa_op_mult_essential_diag(op_, x.Data(), y.Data(), diag_one ? 1 : 0, ctx_->stream, &handled));
  return handled != 0;
}
bool Operator::PrepareFusedStep() const {
  int ok = 0;
  check(pa_op_prepare_fused_step(op_, &ok));
  return ok != 0;
}
void Operator::MultChebyStepEssential(const Vector &x, const ChebyStepArgs &a, bool diag_one) const {
  const pa_cheb_step st{a.sd, a.sr, a.dinv->Data(), a.r0->Data(), a.e_prev ? a.e_prev->Data() : nullptr, a.out->Data(), a.add ? 1 : 0};
  check(pa_op_mult_cheb_step(op_, x.Data(), &st, diag_one ? 1 : 0, ctx_->stream));
}
void Operator::MultResidualEssential(const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0, Vector *d0,
                                     bool diag_one) const {
  check(pa_op_mult_residual(op_, y.Data(), b.Data(), res ? res->Data() : nullptr, dinv ? dinv->Data() : nullptr, c0,
                            d0 ? d0->Data() : nullptr, diag_one ? 1 : 0, ctx_->stream));
}
void Operator::MultSplitStep(const double *x, const double *xg0, const double *xg1, const unsigned long long *sel, double *yg, int n_true,
                             int ess_policy, const pa_split_step &st) const {
  check(pa_op_mult_split_step(op_, x, xg0, xg1, sel, yg, n_true, ess_policy, &st, ctx_->stream));
}
void Operator::Mult2(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1) const {
  check(pa_op_mult2(op_, x0.Data(), x1.Data(), y0.Data(), y1.Data(), ctx_->stream));
}
bool Operator::Mult2EssentialDiag(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1, bool diag_one) const {
  int handled = 0;
  check(pa_op_mult2_essential_diag(op_, x0.Data(), x1.Data(), y0.Data(), y1.Data(), diag_one ? 1 : 0, ctx_->stream, &handled));
  return handled != 0;
}
void Operator::MultEssential(const Vector &x, Vector &y) const {
  check(pa_op_mult_essential(op_, x.Data(), y.Data(), ctx_->stream));
}
}  // namespace ceed

// ---- ParOperator ------------------------------------------------------------------------------
// ---- SumOperator (operator.hpp:132-270) ---------------------------------------------------------
void DiagonalOperator::Mult(const Vector &x, Vector &y) const {
  PA_REQUIRE(x.Size() == d_.Size() && y.Size() == d_.Size(), "size mismatch in DiagonalOperator");
  launch_ew(OpDiagMult{1.0, d_.Data(), x.Data(), y.Data(), 0}, x.Size(), ctx_->stream);
}
void DiagonalOperator::AddMult(const Vector &x, Vector &y, double a) const {
  PA_REQUIRE(x.Size() == d_.Size() && y.Size() == d_.Size(), "size mismatch in DiagonalOperator");
  launch_ew(OpDiagMult{a, d_.Data(), x.Data(), y.Data(), 1}, x.Size(), ctx_->stream);
}
void DiagonalOperator::AssembleDiagonal(Vector &diag) const { linalg::Copy(*ctx_, d_, diag); }

void SumOperator::AddOperator(const Operator &op, double a) {
  StreamGraph::Invalidate();
  PA_REQUIRE(op.Height() == height && op.Width() == width, "Invalid Operator dimensions for BaseSumOperator!");
  ops_.emplace_back(&op, a);
}
void SumOperator::Mult(const Vector &x, Vector &y) const {
  PA_REQUIRE(!ops_.empty(), "empty SumOperator");
  ops_[0].first->Mult(x, y);  // operator.hpp:208-221 (first term written, the others accumulated)
  if (ops_[0].second != 1.0) linalg::AXPBY(*ctx_, 0.0, y, ops_[0].second, y);
  for (size_t k = 1; k < ops_.size(); k++) {
    if (z_.Size() != height) z_.SetSize(height);
    ops_[k].first->Mult(x, z_);
    linalg::AXPY(*ctx_, ops_[k].second, z_, y);
  }
}
void SumOperator::MultTranspose(const Vector &x, Vector &y) const {
  PA_REQUIRE(!ops_.empty(), "empty SumOperator");
  ops_[0].first->MultTranspose(x, y);
  if (ops_[0].second != 1.0) linalg::AXPBY(*ctx_, 0.0, y, ops_[0].second, y);
  for (size_t k = 1; k < ops_.size(); k++) {
    if (z_.Size() != height) z_.SetSize(height);
    ops_[k].first->MultTranspose(x, z_);
    linalg::AXPY(*ctx_, ops_[k].second, z_, y);
  }
}
void SumOperator::AddMultTranspose(const Vector &x, Vector &y, double a) const {
  if (z_.Size() != height) z_.SetSize(height);
  for (const auto &[op, c] : ops_) {
    op->MultTranspose(x, z_);
    linalg::AXPY(*ctx_, a * c, z_, y);
  }
}
bool SumOperator::IsSymmetric() const {
  for (const auto &[op, c] : ops_)
    if (!op->IsSymmetric()) return false;
  return true;
}
void SumOperator::AddMult(const Vector &x, Vector &y, double a) const {
  if (z_.Size() != height) z_.SetSize(height);
  for (const auto &[op, c] : ops_) {
    op->Mult(x, z_);
    linalg::AXPY(*ctx_, a * c, z_, y);
  }
}
void SumOperator::AssembleDiagonal(Vector &diag) const {
  linalg::Fill(*ctx_, diag, 0.0);
  if (z_.Size() != height) z_.SetSize(height);
  for (const auto &[op, c] : ops_) {
    op->AssembleDiagonal(z_);
    linalg::AXPY(*ctx_, c, z_, diag);
  }
}

namespace {
// multi-rank ParOperator::Mult: tx = x with the essential entries zeroed; y = ly with the essential rows set to x or 0
__global__ void k_copy_masked(const double *__restrict__ x, const uint8_t *__restrict__ mask, double *__restrict__ out, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (mask[i] & 1) ? 0.0 : x[i];
}
__global__ void k_copy_fix(const double *__restrict__ ly, const double *__restrict__ x, const uint8_t *__restrict__ mask,
                           const int one, double *__restrict__ y, const int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (mask[i] & 1) ? (one ? x[i] : 0.0) : ly[i];
}
}  // namespace

ParOperator::ParOperator(const Context &ctx, const Operator &A, int n_true, const int32_t *ess_host, int n_ess,
                         DiagonalPolicy policy, const Halo *halo)
    : Operator(n_true, n_true), ctx_(&ctx), A_(&A), halo_(halo), n_true_(n_true), n_local_(A.Height()),
      n_ess_(n_ess), policy_(policy) {
  PA_REQUIRE(A.Height() == A.Width(), "ParOperator needs a square local operator");
  PA_REQUIRE(n_true <= n_local_, "more true dofs than local dofs");
  PA_REQUIRE(halo != nullptr || n_true == n_local_, "local != true dofs requires a halo plan");
  if (halo) halo->Validate(n_true, n_local_);
  for (int i = 0; i < n_ess; i++) PA_REQUIRE(ess_host[i] >= 0 && ess_host[i] < n_true, "essential dof out of range");
  if (n_ess) d_ess_ = pa::dev_upload(ess_host, (size_t)n_ess, ctx.stream);
  if (n_ess) ess_host_.assign(ess_host, ess_host + n_ess);
  lx_.SetSize(n_local_);
  ly_.SetSize(n_local_);
  // One rank (P = identity): let the element kernel read essential entries as zero and write y
  // in place, instead of copying x and y through the L-vectors.
  // (the flagged index tables belong to the local operator: if another wrapper has fused a different list into it,
  // this one masks and fixes its rows outside the kernels instead of overwriting that list)
  if (!halo && n_ess) {
    if (auto *c = dynamic_cast<const ceed::Operator *>(&A)) {
      const int st = pa_op_essential_state(c->Handle(), ess_host, n_ess);
      if (st == 0) const_cast<ceed::Operator *>(c)->SetEssential(ess_host, n_ess);
      if (st >= 0) A_fused_ = c;
    }
  }
  // Peer transport and a local operator that applies to split vectors: no L-vector copies at all (Mult below).  The essential
  // list is fused into the operator's index tables as on one rank (unless another wrapper has fused a different one).
  // (read at every construction: a driver that finds the two forms disagreeing switches the direct one off for what it builds next)
  const bool direct_env = !(std::getenv("PALACE_AMD_HALO_DIRECT") && std::getenv("PALACE_AMD_HALO_DIRECT")[0] == '0');
  if (halo && direct_env && halo->DirectOk(n_true, n_local_)) {
    if (auto *c = dynamic_cast<const ceed::Operator *>(&A)) {
      if (c->SupportsSplit() && c->IsSymmetric()) {
        bool ok = true;
        if (n_ess) {
          const int st = pa_op_essential_state(c->Handle(), ess_host, n_ess);
          if (st == 0) const_cast<ceed::Operator *>(c)->SetEssential(ess_host, n_ess);
          ok = st >= 0;
          split_ess_ = ok;
        }
        if (ok) A_split_ = A_split_avail_ = c;
      }
    } else if (auto *m = dynamic_cast<const CsrOperator *>(&A)) {
      // assembled local operator: this wrapper's essential rows / columns folded into its own copy of the values
      d_csr_bc_ = m->EliminatedValues(d_ess_, n_ess, policy == DiagonalPolicy::DIAG_ONE);
      if (d_csr_bc_) A_csr_split_ = A_csr_split_avail_ = m;
    }
  }
  if (halo && (n_ess || halo->UsesPeerTransport())) {
    // one byte per true dof: bit 1 essential, bit 2 (peer transport) an owned dof other ranks hold as a ghost
    std::vector<uint8_t> mask((size_t)n_true, 0);
    for (int i = 0; i < n_ess; i++) mask[ess_host[i]] = 1;
    for (const int32_t d : halo->SharedOwnedDofs())
      if (d < n_true) mask[d] |= 2;
    d_ess_mask_ = pa::dev_upload(mask.data(), mask.size(), ctx.stream);
  }
  if (halo) {
    // interior element batches can overlap with the exchange of the ghosts (PALACE_AMD_OVERLAP=1).  Off by default: on one GPU
    // with the exchanges redirected to the rank itself (scripts/time_halo_mult.py, the 1/8 slab of the strong-scaling bench)
    // the fork / join of the second stream costs 29 us per apply (104 us against 75 us), more than the transfer of the
    // 0.57 MB interface it could hide behind the interior batches; it pays only for exchanges slower than that.
    static const bool enabled = [] {
      const char *e = std::getenv("PALACE_AMD_OVERLAP");
      return e && e[0] == '1';
    }();
    if (auto *c = dynamic_cast<const ceed::Operator *>(&A); c && enabled) {
      const_cast<ceed::Operator *>(c)->SetInterfaceDofs(halo->InterfaceDofs());
      A_overlap_ = c;
    }
  }
  if (!halo) {
    if (auto *m = dynamic_cast<const CsrOperator *>(&A)) {
      d_csr_bc_ = m->EliminatedValues(d_ess_, n_ess, policy == DiagonalPolicy::DIAG_ONE);
      if (d_csr_bc_) A_csr_ = m;
    }
  }
}
ParOperator::~ParOperator() {
  if (d_ess_mask_) (void)hipFree(d_ess_mask_);
  if (d_ess_) (void)hipFree(d_ess_);
  if (d_csr_bc_) (void)hipFree(d_csr_bc_);
}

bool ParOperator::PrepareChebyStep() const {
  if (A_fused_ && !halo_) return A_fused_->PrepareFusedStep();
  if (A_csr_ && !halo_) return A_csr_->PrepareChebyStep();  // assembled local operator, one rank: the step in the sparse product (csr_op.hip)
  // several ranks, direct form of the peer transport (round 6): the local gather consumes the dofs no other rank shares, the merged
  // P^T kernel the interface dofs (Halo::RestrictAddDirectStep); needs the essential list fused (split_ess_) and the mask
  static const bool halo_step = !(std::getenv("PALACE_AMD_FUSED_STEP_HALO") && std::getenv("PALACE_AMD_FUSED_STEP_HALO")[0] == '0');
  if (halo_step && A_split_ && halo_ && d_ess_mask_ && split_ess_ && halo_->StepOk()) return A_split_->PrepareFusedStep();
  return false;
}
void ParOperator::SplitStep(const Vector &x, const pa_split_step &st0, const HaloStep &hs) const {
  const Context &c = *ctx_;
  pa_split_step st = st0;
  st.iface_mask = d_ess_mask_, st.t_iface = ly_.Data();  // (the L-vector scratch is free in the direct form)
  halo_->SendDirect(x.Data(), d_ess_mask_, c.stream);
  A_split_->MultSplitStep(x.Data(), halo_->GhostIn(0), halo_->GhostIn(1), halo_->GhostInSelector(), halo_->GhostOut(), n_true_,
                          policy_ == DiagonalPolicy::DIAG_ONE ? 1 : 0, st);
  halo_->RestrictAddDirectStep(d_ess_mask_, ly_.Data(), hs, c.stream);
}
void ParOperator::MultChebyStep(const Vector &x, const ChebyStepArgs &a) const {
  if (A_fused_ && !halo_) return A_fused_->MultChebyStepEssential(x, a, policy_ == DiagonalPolicy::DIAG_ONE);
  if (A_csr_ && !halo_) return A_csr_->MultChebyStepValues(d_csr_bc_, x, a);  // (essential rows / columns live in the values)
  PA_REQUIRE(A_split_ && halo_, "MultChebyStep: PrepareChebyStep found no fused form");
  const double *ep = a.e_prev ? a.e_prev->Data() : nullptr;
  SplitStep(x, pa_split_step{1, a.sd, a.sr, a.dinv->Data(), a.r0->Data(), ep, a.out->Data(), a.add ? 1 : 0, nullptr, nullptr, nullptr},
            HaloStep{1, a.sd, a.sr, a.dinv->Data(), a.r0->Data(), x.Data(), ep, a.out->Data(), a.add ? 1 : 0, nullptr});
}
void ParOperator::MultResidual(const Vector &y, const Vector &b, Vector *res, const Vector *dinv, double c0, Vector *d0) const {
  if (A_fused_ && !halo_) return A_fused_->MultResidualEssential(y, b, res, dinv, c0, d0, policy_ == DiagonalPolicy::DIAG_ONE);
  if (A_csr_ && !halo_) return A_csr_->MultResidualValues(d_csr_bc_, y, b, res, dinv, c0, d0);
  PA_REQUIRE(A_split_ && halo_, "MultResidual: PrepareChebyStep found no fused form");
  const double *di = dinv ? dinv->Data() : nullptr;
  double *r = res ? res->Data() : nullptr, *o = d0 ? d0->Data() : nullptr;
  SplitStep(y, pa_split_step{2, 0.0, c0, di, b.Data(), nullptr, o, 0, r, nullptr, nullptr},
            HaloStep{2, 0.0, c0, di, b.Data(), nullptr, nullptr, o, 0, r});
}
void ParOperator::Mult(const Vector &x, Vector &y) const {
  // rap.cpp:195-234.  tx = x, tx[ess] = 0; lx = P tx; ly = A lx; y = P^T ly; y[ess] = x[ess] | 0
  const Context &c = *ctx_;
  if (A_fused_ && x.Data() != y.Data()) {
    if (A_fused_->MultEssentialDiag(x, y, policy_ == DiagonalPolicy::DIAG_ONE)) return;
    if (policy_ == DiagonalPolicy::DIAG_ONE)
      linalg::SetSubVector(c, y, d_ess_, n_ess_, x);
    else
      linalg::SetSubVector(c, y, d_ess_, n_ess_, 0.0);
    return;
  }
  if (A_csr_ && x.Data() != y.Data()) {
    A_csr_->MultValues(d_csr_bc_, x, y);  // essential rows / columns live in this wrapper's copy of the values
    return;
  }
  if (A_split_ && d_ess_mask_ && x.Data() != y.Data()) {
    // peer transport, direct form: masked owned values to the neighbours (the kernel returns when theirs have arrived), the element
    // kernel reads x and the mailbox, the run gather writes y (essential rows fixed) and the ghost rows, those go to their owners
    // and are added to y in place -- five launches, no L-vector
    halo_->SendDirect(x.Data(), d_ess_mask_, c.stream);
    A_split_->MultSplit(x.Data(), halo_->GhostIn(0), halo_->GhostIn(1), halo_->GhostInSelector(), y.Data(), halo_->GhostOut(),
                        n_true_, split_ess_ ? (policy_ == DiagonalPolicy::DIAG_ONE ? 1 : 0) : -1);
    halo_->RestrictAddDirect(d_ess_mask_, y.Data(), c.stream);
    return;
  }
  if (A_csr_split_ && d_ess_mask_ && x.Data() != y.Data()) {  // the same with an assembled local operator
    halo_->SendDirect(x.Data(), d_ess_mask_, c.stream);
    A_csr_split_->MultSplit(d_csr_bc_, x.Data(), halo_->GhostIn(0), halo_->GhostIn(1), halo_->GhostInSelector(), y.Data(),
                            halo_->GhostOut(), n_true_);
    halo_->RestrictAddDirect(d_ess_mask_, y.Data(), c.stream);
    return;
  }
  if (halo_ && halo_->UsesPeerTransport() && d_ess_mask_ && x.Data() != y.Data() && !A_overlap_) {
    // peer transport: the copies x -> lx, ly -> y and the essential-dof handling ride in the exchange kernels
    halo_->ProlongateFused(x.Data(), d_ess_mask_, n_true_, lx_.Data(), c.stream);
    A_->Mult(lx_, ly_);
    halo_->RestrictAddFused(ly_.Data(), x.Data(), d_ess_mask_, policy_ == DiagonalPolicy::DIAG_ONE, n_true_, y.Data(), c.stream);
    return;
  }
  Vector tx(lx_.Data(), n_true_);
  const bool one_launch = d_ess_mask_ && x.Data() != y.Data();
  if (one_launch) {
    hipLaunchKernelGGL(k_copy_masked, dim3((n_true_ + 255) / 256), dim3(256), 0, c.stream, x.Data(), d_ess_mask_, tx.Data(), n_true_);
  } else {
    linalg::Copy(c, x, tx);
    if (n_ess_) linalg::SetSubVector(c, tx, d_ess_, n_ess_, 0.0);  // (before P: the ghost copies of essential dofs must be zero too)
  }
  if (halo_ && A_overlap_ && !StreamGraph::Recording()) {
    // P on a second stream: fork after tx is complete, the apply joins before its interface batches
    Workspace &w = c.Work();
    hipStream_t hs = w.HaloStream();
    PA_HIP(hipEventRecord(w.ReadyEvent(), c.stream));
    PA_HIP(hipStreamWaitEvent(hs, w.ReadyEvent(), 0));
    halo_->Prolongate(lx_.Data(), hs);
    PA_HIP(hipEventRecord(w.DoneEvent(), hs));
    A_overlap_->MultAfter(lx_, ly_, w.DoneEvent());
  } else {
    if (halo_) halo_->Prolongate(lx_.Data(), c.stream);  // owners -> sharers (P)
    A_->Mult(lx_, ly_);
  }
  if (halo_) halo_->RestrictAdd(ly_.Data(), c.stream);  // sharers -> owners, summed (P^T)
  Vector ty(ly_.Data(), n_true_);
  if (one_launch) {
    hipLaunchKernelGGL(k_copy_fix, dim3((n_true_ + 255) / 256), dim3(256), 0, c.stream, ty.Data(), x.Data(), d_ess_mask_,
                       policy_ == DiagonalPolicy::DIAG_ONE ? 1 : 0, y.Data(), n_true_);
    return;
  }
  linalg::Copy(c, ty, y);
  if (n_ess_) {
    if (policy_ == DiagonalPolicy::DIAG_ONE)
      linalg::SetSubVector(c, y, d_ess_, n_ess_, x);
    else
      linalg::SetSubVector(c, y, d_ess_, n_ess_, 0.0);
  }
}

void ParOperator::MultTranspose(const Vector &x, Vector &y) const {
  // rap.cpp:236-275.  ty = x, ty[ess] = 0; ly = P ty; lx = A^T ly; y = P^T lx; y[ess] = x[ess] | 0
  if (A_->IsSymmetric()) return Mult(x, y);
  const Context &c = *ctx_;
  Vector tx(lx_.Data(), n_true_);
  linalg::Copy(c, x, tx);
  if (n_ess_) linalg::SetSubVector(c, tx, d_ess_, n_ess_, 0.0);
  if (halo_) halo_->Prolongate(lx_.Data(), c.stream);
  if (A_csr_)
    throw pa::Error("transpose of an assembled non-symmetric local operator is not available");
  A_->MultTranspose(lx_, ly_);
  if (halo_) halo_->RestrictAdd(ly_.Data(), c.stream);
  Vector ty(ly_.Data(), n_true_);
  linalg::Copy(c, ty, y);
  if (n_ess_) {
    if (policy_ == DiagonalPolicy::DIAG_ONE)
      linalg::SetSubVector(c, y, d_ess_, n_ess_, x);
    else
      linalg::SetSubVector(c, y, d_ess_, n_ess_, 0.0);
  }
}

void ParOperator::AddMultTranspose(const Vector &x, Vector &y, double a) const {
  if (tt_.Size() != n_true_) tt_.SetSize(n_true_);
  MultTranspose(x, tt_);
  linalg::AXPY(*ctx_, a, tt_, y);
}

void ParOperator::Mult2(const Vector &x0, const Vector &x1, Vector &y0, Vector &y1) const {
  const Context &c = *ctx_;
  // two right-hand sides in one pass of the one-shot kernel pay off only where there is no streaming kernel: with it, two
  // separate applies are faster (complex K - w^2 M + i w C apply on 9.95M dofs: 0.86 ms against 1.00 ms)
  if (A_fused_ && !A_fused_->Streams()) {
    const bool one = policy_ == DiagonalPolicy::DIAG_ONE;
    if (!A_fused_->Mult2EssentialDiag(x0, x1, y0, y1, one)) {
      if (one)
        linalg::SetSubVector(c, y0, d_ess_, n_ess_, x0), linalg::SetSubVector(c, y1, d_ess_, n_ess_, x1);
      else
        linalg::SetSubVector(c, y0, d_ess_, n_ess_, 0.0), linalg::SetSubVector(c, y1, d_ess_, n_ess_, 0.0);
    }
    return;
  }
  Mult(x0, y0);
  Mult(x1, y1);
}

void ParOperator::AddMult(const Vector &x, Vector &y, double a) const {
  if (tt_.Size() != n_true_) tt_.SetSize(n_true_);
  Mult(x, tt_);
  linalg::AXPY(*ctx_, a, tt_, y);
}

void ParOperator::EliminateRHS(const Vector &x, Vector &b) const {
  // rap.cpp:56-82: tx = 0, tx[ess] = x[ess]; b -= P^T A P tx (unconstrained A); b[ess] = x[ess] | 0
  const Context &c = *ctx_;
  Vector tx(lx_.Data(), n_true_);
  linalg::Fill(c, tx, 0.0);
  if (n_ess_) linalg::SetSubVector(c, tx, d_ess_, n_ess_, x);
  if (halo_) halo_->Prolongate(lx_.Data(), c.stream);
  else if (n_local_ > n_true_)
    PA_HIP(hipMemsetAsync(lx_.Data() + n_true_, 0, sizeof(double) * (size_t)(n_local_ - n_true_), c.stream));
  A_->Mult(lx_, ly_);
  if (halo_) halo_->RestrictAdd(ly_.Data(), c.stream);
  Vector ty(ly_.Data(), n_true_);
  linalg::AXPY(c, -1.0, ty, b);
  if (n_ess_) {
    if (policy_ == DiagonalPolicy::DIAG_ONE)
      linalg::SetSubVector(c, b, d_ess_, n_ess_, x);
    else
      linalg::SetSubVector(c, b, d_ess_, n_ess_, 0.0);
  }
}

void ParOperator::AssembleDiagonal(Vector &diag) const {
  // rap.cpp:154-193 (conforming meshes: |P|^T = P^T)
  const Context &c = *ctx_;
  A_->AssembleDiagonal(ly_);
  if (halo_) halo_->RestrictAdd(ly_.Data(), c.stream);
  Vector ty(ly_.Data(), n_true_);
  linalg::Copy(c, ty, diag);
  if (n_ess_) linalg::SetSubVector(c, diag, d_ess_, n_ess_, policy_ == DiagonalPolicy::DIAG_ONE ? 1.0 : 0.0);
}

// ---- smoothers --------------------------------------------------------------------------------
void JacobiSmoother::SetOperator(const Operator &op) {
  StreamGraph::Invalidate();
  A_ = &op, height = op.Height(), width = op.Width();
  dinv_.SetSize(height);
  op.AssembleDiagonal(dinv_);
  linalg::Reciprocal(*ctx_, dinv_);
}
void JacobiSmoother::Mult(const Vector &x, Vector &y) const {
  // jacobi.cpp:74-104 with zero initial guess: y = D^{-1} x
  linalg::ChebyOrder0(*ctx_, 1.0, dinv_, x, y);
}

void ChebyshevSmoother::SetOperator(const Operator &op) {
  StreamGraph::Invalidate();
  // chebyshev.cpp:169-188 (4th kind) / :232-257 (1st kind)
  A_ = &op, height = op.Height(), width = op.Width();
  d_.SetSize(height), dinv_.SetSize(height), r_.SetSize(height), t_.SetSize(height), w_.SetSize(height);
  op.AssembleDiagonal(dinv_);
  linalg::Reciprocal(*ctx_, dinv_);
  lambda_max_ = sf_max_ * linalg::SpectralNorm(*ctx_, op, dinv_);
  PA_REQUIRE(lambda_max_ > 0.0, "Encountered zero maximum eigenvalue in Chebyshev smoother!");
  if (!fourth_kind_ && sf_min_ <= 0.0) sf_min_ = 1.69 / (std::pow(order_, 1.68) + 2.11 * order_ + 1.98);
  // (read at every set-up: PALACE_AMD_FUSED_STEP=0 keeps the apply + vector kernel pair, for A / B runs in one process)
  const char *fe = std::getenv("PALACE_AMD_FUSED_STEP");
  fused_step_ = !(fe && fe[0] == '0') && order_ > 1 && op.PrepareChebyStep();
}
void ChebyshevSmoother::Mult(const Vector &x, Vector &y) const { Mult2(x, y, r_); }
void ChebyshevSmoother::Mult2(const Vector &x, Vector &y, Vector &r) const {
  // chebyshev.cpp:160-220 (4th kind) and :222-293 (1st kind).  The reference's recurrence -- d_0 = c_0 D^-1 r; then for every
  // further order  y += d, r -= A d, d = sd_k d + sr_k D^-1 r; finally y += d -- is carried here by the accumulated correction
  // e_k = d_0 + ... + d_{k-1} (d_{k-1} = e_k - e_{k-1}, r_k = r_0 - A e_k): the same polynomial, but r_0 is only read (with a
  // zero guess it is the right-hand side x itself: no copy), y is written once, and a step moves 48 instead of 64 bytes per entry
  // (DESIGN.md 3.7).  PALACE_AMD_CHEBY_FORM=d selects the literal form.
  const Context &c = *ctx_;
  static const bool literal = [] {
    const char *e = std::getenv("PALACE_AMD_CHEBY_FORM");
    return e && e[0] == 'd';
  }();
  const double lmax = lambda_max_, lmin = sf_min_ * lmax;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  // coefficients of order k (k = 0: the scale of the first direction)
  auto first = [&]() { return fourth_kind_ ? 4.0 / (3.0 * lmax) : 1.0 / theta; };
  for (int it = 0; it < pc_it_; it++) {
    const bool zero = !(initial_guess || it > 0);
    if (literal) {
      if (!zero) {
        A_->Mult(y, r);
        linalg::AXPBY(c, 1.0, x, -1.0, r);
      } else {
        linalg::Copy(c, x, r);
        linalg::Fill(c, y, 0.0);
      }
      linalg::ChebyOrder0(c, first(), dinv_, r, d_);
      double rhop = delta / theta;
      for (int k = 1; k < order_; k++) {
        double sd, sr;
        if (fourth_kind_) {  // chebyshev.cpp:204-218
          sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / ((2.0 * k + 3.0) * lmax);
        } else {  // chebyshev.cpp:275-291
          const double rho = 1.0 / (2.0 * theta / delta - rhop);
          sd = rho * rhop, sr = 2.0 * rho / delta, rhop = rho;
        }
        A_->Mult(d_, t_);
        linalg::ChebyStep(c, sd, sr, dinv_, t_, r, d_, y);  // y += d; r -= A d; d = sd d + sr D^-1 r
      }
      linalg::AXPY(c, 1.0, d_, y);
      continue;
    }
    const Vector *r0 = &x;
    bool have_e1 = false;
    if (!zero && fused_step_ && order_ > 1 && r.Data() != x.Data()) {
      // r_0 = x - A y and e_1 = c_0 D^-1 r_0 in the epilogue of the operator's E^T (A y is not stored)
      A_->MultResidual(y, x, &r, &dinv_, first(), &d_);
      r0 = &r, have_e1 = true;
    } else if (!zero) {  // r_0 = x - A y
      A_->Mult(y, r);
      linalg::AXPBY(c, 1.0, x, -1.0, r);
      r0 = &r;
    }
    if (order_ <= 1) {  // y (+)= c_0 D^-1 r_0
      if (zero) {
        linalg::ChebyOrder0(c, first(), dinv_, *r0, y);
      } else {
        linalg::ChebyOrder0(c, first(), dinv_, *r0, d_);
        linalg::AXPY(c, 1.0, d_, y);
      }
      continue;
    }
    // e_k and the buffer e_{k+1} goes to (zero guess: the caller's work vector r is free for that -- unless it is x itself)
    Vector *ek = &d_, *ep = (zero && r.Data() != x.Data()) ? &r : &w_;
    if (!have_e1) linalg::ChebyOrder0(c, first(), dinv_, *r0, *ek);  // e_1 = d_0
    double rhop = delta / theta;
    for (int k = 1; k < order_; k++) {
      double sd, sr;
      if (fourth_kind_) {
        sd = (2.0 * k - 1.0) / (2.0 * k + 3.0), sr = (8.0 * k + 4.0) / ((2.0 * k + 3.0) * lmax);
      } else {
        const double rho = 1.0 / (2.0 * theta / delta - rhop);
        sd = rho * rhop, sr = 2.0 * rho / delta, rhop = rho;
      }
      const bool last = k == order_ - 1;
      // e_{k+1} = e_k + sd (e_k - e_{k-1}) + sr D^-1 (r_0 - A e_k); the last one goes (is added) to y
      if (fused_step_) {  // ... inside the operator's E^T: A e_k is consumed where it is produced (round 6)
        A_->MultChebyStep(*ek, Operator::ChebyStepArgs{sd, sr, &dinv_, r0, k == 1 ? nullptr : ep, last ? &y : ep, last && !zero});
      } else {
        A_->Mult(*ek, t_);
        linalg::ChebyStep3(c, sd, sr, dinv_, t_, *r0, *ek, k == 1 ? nullptr : ep, last ? y : *ep, last && !zero);
      }
      if (!last) std::swap(ek, ep);
    }
  }
}

// ---- Hiptmair smoother (distrelaxation.cpp) -----------------------------------------------------------
DistRelaxationSmoother::DistRelaxationSmoother(const Context &ctx, const Operator &G, int smooth_it, int cheby_smooth_it,
                                               int cheby_order, double sf_max, double sf_min, bool fourth)
    : ctx_(&ctx), pc_it_(smooth_it), G_(&G) {
  B_ = std::make_unique<ChebyshevSmoother>(ctx, cheby_smooth_it, cheby_order, sf_max, fourth, sf_min);
  B_G_ = std::make_unique<ChebyshevSmoother>(ctx, cheby_smooth_it, cheby_order, sf_max, fourth, sf_min);
  B_G_->SetInitialGuess(false);
}
void DistRelaxationSmoother::SetOperators(const Operator &op, const ParOperator &op_G) {
  StreamGraph::Invalidate();
  PA_REQUIRE(op.Height() == G_->Height() && op.Width() == G_->Height() && op_G.Height() == G_->Width() &&
                 op_G.Width() == G_->Width(),
             "Invalid operator sizes for DistRelaxationSmoother!");
  A_ = &op, A_G_ = &op_G;
  x_G_.SetSize(op_G.Height()), y_G_.SetSize(op_G.Height()), r_G_.SetSize(op_G.Height());
  t_.SetSize(op.Height());
  B_->SetOperator(op);
  B_G_->SetOperator(op_G);
  height = op.Height(), width = op.Width();
}
void DistRelaxationSmoother::Mult(const Vector &x, Vector &y) const { Mult2(x, y, t_); }
void DistRelaxationSmoother::Mult2(const Vector &x, Vector &y, Vector &r) const {
  const Context &c = *ctx_;
  for (int it = 0; it < pc_it_; it++) {
    // y = y + B (x - A y)
    B_->SetInitialGuess(initial_guess || it > 0);
    B_->Mult2(x, y, r);
    // y = y + G B_G G^T (x - A y)
    if (B_->FusedStep() && r.Data() != x.Data()) {  // (the residual out of the operator's E^T epilogue, round 6)
      A_->MultResidual(y, x, &r);
    } else {
      A_->Mult(y, r);
      linalg::AXPBY(c, 1.0, x, -1.0, r);
    }
    G_->MultTranspose(r, x_G_);
    if (A_G_->NumEssentialTrueDofs())
      linalg::SetSubVector(c, x_G_, A_G_->GetEssentialTrueDofs(), A_G_->NumEssentialTrueDofs(), 0.0);
    B_G_->Mult2(x_G_, y_G_, r_G_);
    G_->Mult(y_G_, r);  // RealAddMult(G, y_G, y)
    linalg::AXPY(c, 1.0, r, y);
  }
}
void DistRelaxationSmoother::MultTranspose2(const Vector &x, Vector &y, Vector &r) const {
  const Context &c = *ctx_;
  B_->SetInitialGuess(true);
  for (int it = 0; it < pc_it_; it++) {
    // y = y + G B_G^T G^T (x - A y)
    if ((initial_guess || it > 0) && B_->FusedStep() && r.Data() != x.Data()) {
      A_->MultResidual(y, x, &r);
      G_->MultTranspose(r, x_G_);
    } else if (initial_guess || it > 0) {
      A_->Mult(y, r);
      linalg::AXPBY(c, 1.0, x, -1.0, r);
      G_->MultTranspose(r, x_G_);
    } else {
      linalg::Fill(c, y, 0.0);
      G_->MultTranspose(x, x_G_);
    }
    if (A_G_->NumEssentialTrueDofs())
      linalg::SetSubVector(c, x_G_, A_G_->GetEssentialTrueDofs(), A_G_->NumEssentialTrueDofs(), 0.0);
    B_G_->MultTranspose2(x_G_, y_G_, r_G_);
    G_->Mult(y_G_, r);
    linalg::AXPY(c, 1.0, r, y);
    // y = y + B^T (x - A y)
    B_->MultTranspose2(x, y, r);
  }
}

// ---- PCG (iterative.cpp:360-486) ----------------------------------------------------------------
struct CgSolver::DeviceState {
  double *d_st = nullptr;    // the scalars (CgSlot)
  double *h_ring = nullptr;  // pinned snapshots of d_st, one slot per iteration in flight
  int ring = 0;
  std::vector<hipEvent_t> ev;
  StreamGraph graph;     // one iteration
  bool pending = false;  // lookahead < 0: the final snapshot has been enqueued but not read
  bool deferred = false; // the last solve ran in the never-wait form: replays of an enclosing recording refresh the snapshot
                         // without running this host code, so the statistics are (re)read whenever someone asks
  ~DeviceState() {
    if (d_st) (void)hipFree(d_st);
    if (h_ring) (void)hipHostFree(h_ring);
    for (auto e : ev) (void)hipEventDestroy(e);
  }
  void Setup(int slots) {
    if (!d_st) d_st = pa::dev_alloc<double>(CG_NSLOT);
    if (slots > ring) {
      if (h_ring) PA_HIP(hipHostFree(h_ring));
      PA_HIP(hipHostMalloc(reinterpret_cast<void **>(&h_ring), (size_t)slots * CG_NSLOT * sizeof(double), hipHostMallocDefault));
      while ((int)ev.size() < slots) {
        hipEvent_t e;
        PA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev.push_back(e);
      }
      ring = slots;
    }
  }
  const double *Slot(int k) const { return h_ring + (size_t)k * CG_NSLOT; }
  void Snapshot(int k, hipStream_t s, bool with_event) {
    PA_HIP(hipMemcpyAsync(h_ring + (size_t)k * CG_NSLOT, d_st, CG_NSLOT * sizeof(double), hipMemcpyDeviceToHost, s));
    if (with_event) PA_HIP(hipEventRecord(ev[k], s));
  }
};

namespace {
// (x, y) into the scalars of the recurrence: local two-stage reduction, all-reduce across ranks, scalar update
template <int STEP>
void cg_dot(const Context &c, const Vector &x, const Vector &y, double *st, const CgTol &tol) {
  const Scratch s = scratch(c);
  const bool wide = ((bits(x.Data()) | bits(y.Data())) & 15) == 0 && x.Size() >= 2;
  const int nb = grid_for(wide ? (x.Size() + 1) / 2 : std::max(x.Size(), 1));
  if (wide)
    hipLaunchKernelGGL(k_dot_partial<2>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), y.Data(), (long long)x.Size(),
                       s.d_partial);
  else
    hipLaunchKernelGGL(k_dot_partial<1>, dim3(nb), dim3(kBlock), 0, c.stream, x.Data(), y.Data(), (long long)x.Size(),
                       s.d_partial);
  if (c.comm && c.comm->Size() > 1) {
    hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(kBlock), 0, c.stream, s.d_partial, nb, st + CG_TMP);
    c.comm->AllReduceSum(st + CG_TMP, 1, c.stream);  // Mpi::GlobalSum, no host in the loop
    hipLaunchKernelGGL(k_cg_scalar<STEP>, dim3(1), dim3(1), 0, c.stream, st, tol);
  } else {
    hipLaunchKernelGGL(k_cg_dot_final<STEP>, dim3(1), dim3(kBlock), 0, c.stream, s.d_partial, nb, st, tol);
  }
  PA_HIP(hipGetLastError());
}
template <class Op>
void launch_ew_cg(const Op &op, long long n, const double *st, hipStream_t stream) {
  if (n <= 0) return;
  if ((op.align() & 15) == 0 && n >= 2)
    hipLaunchKernelGGL((k_ew_cg<2, Op>), dim3(grid_full((n + 1) / 2)), dim3(kBlock), 0, stream, op, n, st);
  else
    hipLaunchKernelGGL((k_ew_cg<1, Op>), dim3(grid_full(n)), dim3(kBlock), 0, stream, op, n, st);
  PA_HIP(hipGetLastError());
}
bool cg_host_scalars_env() {
  static const bool v = [] {
    const char *e = std::getenv("PALACE_AMD_CG_HOST");
    return e && e[0] == '1';
  }();
  return v;
}
}  // namespace

CgSolver::CgSolver(const Context &ctx, int print) : IterativeSolver(ctx, print) {}
CgSolver::~CgSolver() = default;
void CgSolver::SetOperator(const Operator &op) {
  IterativeSolver::SetOperator(op);
  if (dev_) dev_->graph.Reset();
}
void CgSolver::SetPreconditioner(const Solver &pc) {
  IterativeSolver::SetPreconditioner(pc);
  if (dev_) dev_->graph.Reset();
}

void CgSolver::Mult(const Vector &b, Vector &x) const {
  if (host_scalars_ || cg_host_scalars_env())
    MultHost(b, x);
  else
    MultDevice(b, x);
}

void CgSolver::Finish() const {
  if (!dev_ || !(dev_->pending || dev_->deferred)) return;
  StreamGraph::RequireNotRecording("CgSolver statistics");
  PA_HIP(hipStreamSynchronize(ctx_->stream));
  if (ctx_->comm) ctx_->comm->PeerCheckNow();
  dev_->pending = false;
  const double *st = dev_->Slot(0);
  initial_res_ = st[CG_INIT], final_res_ = st[CG_RES], final_it_ = (int)st[CG_IT];
  converged_ = st[CG_BAD] == 0.0 && st[CG_RES] < st[CG_EPS];
  // the solve ran without the host looking (inside a recorded sequence): raise now what CheckDot would have raised then
  PA_REQUIRE(st[CG_BAD] != 1.0, "PCG preconditioner is not positive definite: (Br, r) not finite");
  PA_REQUIRE(st[CG_BAD] != 2.0, "PCG operator is not positive definite: (Ap, p) not finite");
}

void CgSolver::MultDevice(const Vector &b, Vector &x) const {
  const Context &c = *ctx_;
  PA_REQUIRE(A_, "Operator must be set for CgSolver::Mult!");
  const int n = A_->Height();
  r_.SetSize(n), z_.SetSize(n), p_.SetSize(n);
  if (!dev_) dev_ = std::make_unique<DeviceState>();
  DeviceState &d = *dev_;
  const int L = StreamGraph::Recording() ? -1 : lookahead_;  // inside a recorded sequence: never wait
  d.Setup(L >= 0 ? L + 2 : 1);
  d.pending = false, d.deferred = false;
  double *st = d.d_st;
  const CgTol tol{rel_tol_, abs_tol_, initial_guess ? (B_ ? 1 : 2) : 0};
  auto precond = [&](const Vector &u, Vector &v) {
    PhaseRange range("Preconditioner");  // iterative.cpp:247
    if (B_) B_->Mult(u, v); else linalg::Copy(c, u, v);
  };
  auto check = [&](const double *h) {
    PA_REQUIRE(h[CG_BAD] != 1.0, "PCG preconditioner is not positive definite: (Br, r) not finite");
    PA_REQUIRE(h[CG_BAD] != 2.0, "PCG operator is not positive definite: (Ap, p) not finite");
  };
  // iterative.cpp:375-421
  if (initial_guess) {
    A_->Mult(x, r_);
    linalg::AXPBY(c, 1.0, b, -1.0, r_);
    if (B_) {
      B_->Mult(b, p_);
      cg_dot<CG_STEP_RHS>(c, p_, b, st, tol);
    } else {
      cg_dot<CG_STEP_RHS>(c, b, b, st, tol);
    }
  } else {
    linalg::Copy(c, b, r_);
    linalg::Fill(c, x, 0.0);
  }
  precond(r_, z_);
  cg_dot<CG_STEP_START>(c, z_, r_, st, tol);
  linalg::Fill(c, p_, 0.0);  // beta_prev = beta at the start: p = z + 1 * 0
  // one iteration (iterative.cpp:432-474); the same launches for every `it`
  auto iteration = [&]() {
    launch_ew_cg(OpCgDirDev{{1.0, z_.Data(), 1.0, p_.Data()}}, n, st, c.stream);
    A_->Mult(p_, z_);
    cg_dot<CG_STEP_DENOM>(c, z_, p_, st, tol);
    launch_ew_cg(OpCgUpdateDev{{0.0, p_.Data(), z_.Data(), x.Data(), r_.Data()}}, n, st, c.stream);
    precond(r_, z_);
    cg_dot<CG_STEP_BETA>(c, z_, r_, st, tol);
  };
  const std::vector<const void *> key{x.Data(), A_, B_};
  if (L < 0) {
    // fixed work, nothing read back: the stop flag freezes x once the tolerance is met
    for (int it = 0; it < max_it_; it++) d.graph.Run(c, key, iteration);
    d.Snapshot(0, c.stream, false);
    d.pending = true, d.deferred = true;
    return;
  }
  const int R = d.ring;
  d.Snapshot(0, c.stream, true);
  PA_HIP(hipEventSynchronize(d.ev[0]));
  if (c.comm) c.comm->PeerCheckNow();
  check(d.Slot(0));
  int last = 0;  // ring slot of the newest snapshot
  bool stop = d.Slot(0)[CG_STOP] != 0.0;
  int it = 0;
  for (; it < max_it_ && !stop; it++) {
    d.graph.Run(c, key, iteration);
    last = (it + 1) % R;
    d.Snapshot(last, c.stream, true);
    const int j = it - L;  // the newest iteration the host waits for
    if (j >= 0) {
      const int sj = (j + 1) % R;
      PA_HIP(hipEventSynchronize(d.ev[sj]));
      if (c.comm) c.comm->PeerCheckNow();
      const double *h = d.Slot(sj);
      check(h);
      if (print_ > 1) std::printf("  %3d KSP residual norm ||r||_B = %.6e\n", j + 1, h[CG_RES]);
      stop = h[CG_STOP] != 0.0;
    }
  }
  PA_HIP(hipEventSynchronize(d.ev[last]));
  if (c.comm) c.comm->PeerCheckNow();
  const double *h = d.Slot(last);
  check(h);
  initial_res_ = h[CG_INIT], final_res_ = h[CG_RES], final_it_ = (int)h[CG_IT];
  converged_ = h[CG_RES] < h[CG_EPS];
  if (B_) B_->CheckStatus();  // deferred failures of nested solvers (the coarse PCG inside the recorded V-cycle)
  if (print_ > 0)
    std::printf("  PCG solver %s in %d iterations (res %.3e, initial %.3e)\n",
                converged_ ? "converged" : "did NOT converge", final_it_, final_res_, initial_res_);
}

void CgSolver::MultHost(const Vector &b, Vector &x) const {
  const Context &c = *ctx_;
  PA_REQUIRE(A_, "Operator must be set for CgSolver::Mult!");
  const int n = A_->Height();
  r_.SetSize(n), z_.SetSize(n), p_.SetSize(n);
  double beta, beta_prev = 0.0, alpha, denom, res, eps;
  if (initial_guess) {
    A_->Mult(x, r_);
    linalg::AXPBY(c, 1.0, b, -1.0, r_);
  } else {
    linalg::Copy(c, b, r_);
    linalg::Fill(c, x, 0.0);
  }
  if (B_) B_->Mult(r_, z_); else linalg::Copy(c, r_, z_);
  beta = linalg::Dot(c, z_, r_);
  PA_REQUIRE(std::isfinite(beta), "PCG preconditioner is not positive definite: (Br, r) not finite");
  res = std::sqrt(std::abs(beta));
  if (initial_guess) {
    double beta_rhs;
    if (B_) {
      B_->Mult(b, p_);
      beta_rhs = linalg::Dot(c, p_, b);
    } else {
      beta_rhs = linalg::Norml2(c, b);
    }
    initial_res_ = std::sqrt(std::abs(beta_rhs));
  } else {
    initial_res_ = res;
  }
  eps = std::max(rel_tol_ * initial_res_, abs_tol_);
  converged_ = (res < eps);
  int it = 0;
  for (; it < max_it_ && !converged_; it++) {
    if (print_ > 1) std::printf("  %3d KSP residual norm ||r||_B = %.6e\n", it, res);
    if (!it)
      linalg::Copy(c, z_, p_);
    else
      linalg::AXPBY(c, 1.0, z_, beta / beta_prev, p_);
    A_->Mult(p_, z_);
    denom = linalg::Dot(c, z_, p_);
    PA_REQUIRE(std::isfinite(denom), "PCG operator is not positive definite: (Ap, p) not finite");
    alpha = beta / denom;
    linalg::CgUpdate(c, alpha, p_, z_, x, r_);  // x += alpha p; r -= alpha z
    beta_prev = beta;
    if (B_) B_->Mult(r_, z_); else linalg::Copy(c, r_, z_);
    beta = linalg::Dot(c, z_, r_);
    PA_REQUIRE(std::isfinite(beta), "PCG preconditioner is not positive definite: (Br, r) not finite");
    res = std::sqrt(std::abs(beta));
    converged_ = (res < eps);
  }
  if (print_ > 0)
    std::printf("  PCG solver %s in %d iterations (res %.3e, initial %.3e)\n",
                converged_ ? "converged" : "did NOT converge", it, res, initial_res_);
  final_res_ = res, final_it_ = it;
}

// ---- GMRES / FGMRES (iterative.cpp:543-871): the shared implementation (krylov_impl.hpp) on real device vectors -------
namespace {
struct RealKrylovOps {
  using Vec = Vector;
  using Scalar = double;
  const Context &c;
  const Operator *A_;
  const Solver *B_;
  int n;
  void Ensure(Vec &v) const {
    if (v.Size() != n) v.SetSize(n);
  }
  void A(const Vec &x, Vec &y) const { A_->Mult(x, y); }
  bool HasB() const { return B_ != nullptr; }
  void B(const Vec &x, Vec &y) const {
    PhaseRange range("Preconditioner");  // iterative.cpp:247
    B_->Mult(x, y);
  }
  void Copy(const Vec &x, Vec &y) const { linalg::Copy(c, x, y); }
  void Zero(Vec &x) const { linalg::Fill(c, x, 0.0); }
  void BMinus(const Vec &b, Vec &r) const { linalg::AXPBY(c, 1.0, b, -1.0, r); }
  void Axpy(double a, const Vec &x, Vec &y) const { linalg::AXPY(c, a, x, y); }
  void Scale(double s, Vec &x) const { linalg::Scale(c, s, x); }
  double Norm(const Vec &x) const { return linalg::Norml2(c, x); }
  double Orthonormalize(Orthogonalization kind, const std::vector<Vec> &V, Vec &w, double *H, int m) const {
    return linalg::OrthonormalizeColumn(c, kind, V, w, H, m);
  }
};
}  // namespace

void GmresSolver::Mult(const Vector &b, Vector &x) const {
  PA_REQUIRE(A_, "Operator must be set for GmresSolver::Mult!");
  RealKrylovOps ops{*ctx_, A_, B_, A_->Height()};
  krylov::Params p;
  p.rel_tol = rel_tol_, p.abs_tol = abs_tol_, p.max_it = max_it_, p.max_dim = max_dim_, p.print = print_;
  p.flexible = flexible_, p.initial_guess = initial_guess;
  p.pc_side = pc_side_ == PreconditionerSide::RIGHT ? krylov::PreconditionerSide::RIGHT : krylov::PreconditionerSide::LEFT;
  p.orthog = orthog_, p.name = flexible_ ? "FGMRES" : "GMRES";
  krylov::Result res;
  krylov::GmresMult(ops, p, b, x, V_, Z_, r_, res);
  converged_ = res.converged, initial_res_ = res.initial_res, final_res_ = res.final_res, final_it_ = res.final_it;
  if (B_) B_->CheckStatus();
}

// ---- replicated solve of a small global problem ---------------------------------------------------------------------------
namespace {
__global__ void k_scatter_rows(const int n, const int32_t *__restrict__ rows, const double *__restrict__ sgn,
                               const double *__restrict__ x, double *__restrict__ g) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[rows[i]] = sgn ? sgn[i] * x[i] : x[i];
}
__global__ void k_gather_rows(const int n, const int32_t *__restrict__ rows, const double *__restrict__ sgn,
                              const double *__restrict__ g, double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sgn ? sgn[i] * g[rows[i]] : g[rows[i]];
}
}  // namespace
ReplicatedSolver::ReplicatedSolver(const Context &ctx, const Halo &gather, const Solver &inner, const int32_t *mine_host, int n_true,
                                   int n_global, const double *sign_host)
    : ctx_(&ctx), gather_(&gather), inner_(&inner), n_true_(n_true), n_global_(n_global) {
  PA_REQUIRE(n_true >= 0 && n_true <= n_global && inner.Height() == n_global, "replicated solver: sizes do not match");
  for (int i = 0; i < n_true; i++) PA_REQUIRE(mine_host[i] >= 0 && mine_host[i] < n_global, "global dof number out of range");
  height = width = n_true;
  if (n_true) d_mine_ = pa::dev_upload(mine_host, (size_t)n_true, ctx.stream);
  if (n_true && sign_host) d_sign_ = pa::dev_upload(sign_host, (size_t)n_true, ctx.stream);
  gx_.SetSize(n_global), gy_.SetSize(n_global);
  linalg::Fill(ctx, gx_, 0.0);
}
ReplicatedSolver::~ReplicatedSolver() {
  if (d_mine_) (void)hipFree(d_mine_);
  if (d_sign_) (void)hipFree(d_sign_);
}
void ReplicatedSolver::Mult(const Vector &x, Vector &y) const {
  PhaseRange range("Coarse Solve / Replicated");
  const Context &c = *ctx_;
  PA_REQUIRE(x.Size() == n_true_ && y.Size() == n_true_, "size mismatch in ReplicatedSolver");
  if (n_true_)
    hipLaunchKernelGGL(k_scatter_rows, dim3((n_true_ + 255) / 256), dim3(256), 0, c.stream, n_true_, d_mine_, d_sign_, x.Data(),
                       gx_.Data());
  gather_->Prolongate(gx_.Data(), c.stream);  // owners -> everybody: the global right-hand side on every rank
  const_cast<Solver *>(inner_)->SetInitialGuess(false);
  inner_->Mult(gx_, gy_);
  if (n_true_)
    hipLaunchKernelGGL(k_gather_rows, dim3((n_true_ + 255) / 256), dim3(256), 0, c.stream, n_true_, d_mine_, d_sign_, gy_.Data(),
                       y.Data());
  PA_HIP(hipGetLastError());
}

// ---- geometric multigrid (gmg.cpp) ---------------------------------------------------------------
GeometricMultigridSolver::GeometricMultigridSolver(const Context &ctx, std::unique_ptr<Solver> &&coarse_solver,
                                                   const std::vector<const Operator *> &P, int cycle_it, int smooth_it,
                                                   int cheby_order, double cheby_sf_max, double cheby_sf_min,
                                                   bool cheby_4th_kind, const std::vector<const Operator *> *G)
    : ctx_(&ctx), pc_it_(cycle_it), P_(P), A_(P.size() + 1), B_(P.size() + 1), X_(P.size() + 1), Y_(P.size() + 1),
      R_(P.size() + 1) {
  PA_REQUIRE(!G || G->size() == B_.size(),
             "Invalid input for distributive relaxation smoother auxiliary space transfer operators!");
  B_[0] = std::move(coarse_solver);
  // a Krylov coarse solve inside the cycle must not stall the stream: fixed work, statistics left on the device
  if (auto *cg = dynamic_cast<CgSolver *>(B_[0].get())) cg->SetLookahead(-1);
  for (size_t l = 1; l < B_.size(); l++) {
    if (G)  // gmg.cpp:41-47: cheby_smooth_it = 1 inside the distributive relaxation
      B_[l] = std::make_unique<DistRelaxationSmoother>(ctx, *(*G)[l], smooth_it, 1, cheby_order, cheby_sf_max,
                                                       cheby_sf_min, cheby_4th_kind);
    else
      B_[l] = std::make_unique<ChebyshevSmoother>(ctx, smooth_it, cheby_order, cheby_sf_max, cheby_4th_kind,
                                                  cheby_sf_min);
  }
}

void GeometricMultigridSolver::SetOperators(const std::vector<const ParOperator *> &ops,
                                            const std::vector<const ParOperator *> *aux_ops) {
  StreamGraph::Invalidate();
  PA_REQUIRE(ops.size() == A_.size(), "Invalid number of levels for operators in multigrid solver setup!");
  for (size_t l = 0; l < ops.size(); l++) {
    A_[l] = ops[l];
    PA_REQUIRE(A_[l]->Width() == A_[l]->Height(), "Invalid operator sizes for GeometricMultigridSolver!");
    if (l + 1 < ops.size()) PA_REQUIRE(A_[l]->Height() == P_[l]->Width(), "Prolongation / operator size mismatch");
    if (l > 0) PA_REQUIRE(A_[l]->Height() == P_[l - 1]->Height(), "Prolongation / operator size mismatch");
    if (auto *dist = dynamic_cast<DistRelaxationSmoother *>(B_[l].get())) {
      PA_REQUIRE(aux_ops && aux_ops->size() == ops.size(),
                 "Distributive relaxation smoother relies on both primary space and auxiliary space operators!");
      dist->SetOperators(*A_[l], *(*aux_ops)[l]);
    } else {
      B_[l]->SetOperator(*A_[l]);
    }
    X_[l].SetSize(A_[l]->Height()), Y_[l].SetSize(A_[l]->Height()), R_[l].SetSize(A_[l]->Height());
  }
  fused_res_.assign(ops.size(), 0);
  {
    const char *fe = std::getenv("PALACE_AMD_FUSED_STEP");
    for (size_t l = 1; l < ops.size(); l++) fused_res_[l] = !(fe && fe[0] == '0') && A_[l]->PrepareChebyStep();
  }
  height = width = ops.back()->Height();
  graph_.Reset(), graph_alias_.Reset();
  last_x_ = last_y_ = nullptr;
}

void GeometricMultigridSolver::Mult(const Vector &x, Vector &y) const {
  const int L = (int)A_.size();
  // A caller that comes back with the same pair of vectors (PCG: r, z at every iteration) gets the cycle on those vectors
  // themselves -- x is only read by the cycle, y is written by the first smoother: two copies of the finest vectors less per
  // application -- with its own recording.  Everybody else (GMRES / FGMRES hand over a different basis vector every time) goes
  // through the solver's own vectors, so that ONE recording serves every (x, y).
  const bool same_pair = x.Data() == last_x_ && y.Data() == last_y_ && x.Data() != y.Data();
  last_x_ = x.Data(), last_y_ = y.Data();
  if (same_pair) {
    X_[L - 1].MakeRef(const_cast<double *>(x.Data()), x.Size());
    Y_[L - 1].MakeRef(y.Data(), y.Size());
    graph_alias_.Run(*ctx_, {this, x.Data(), y.Data()}, [&] {
      for (int it = 0; it < pc_it_; it++) VCycle(L - 1, it > 0);
    });
    return;
  }
  if (Xown_.Size() != x.Size()) Xown_.SetSize(x.Size()), Yown_.SetSize(x.Size());
  linalg::Copy(*ctx_, x, Xown_);
  X_[L - 1].MakeRef(Xown_.Data(), x.Size());
  Y_[L - 1].MakeRef(Yown_.Data(), x.Size());
  graph_.Run(*ctx_, {this}, [&] {
    for (int it = 0; it < pc_it_; it++) VCycle(L - 1, it > 0);
  });
  linalg::Copy(*ctx_, Yown_, y);
}

void GeometricMultigridSolver::VCycle(int l, bool initial_guess) const {
  // gmg.cpp:171-205
  const Context &c = *ctx_;
  B_[l]->SetInitialGuess(initial_guess);
  if (l == 0) {
    PhaseRange range("Coarse Solve");  // gmg.cpp:180
    B_[l]->Mult(X_[l], Y_[l]);
    return;
  }
  B_[l]->Mult2(X_[l], Y_[l], R_[l]);
  if (fused_res_[l]) {  // r = x - A y in the epilogue of the operator's E^T (round 6)
    A_[l]->MultResidual(Y_[l], X_[l], &R_[l]);
  } else {
    A_[l]->Mult(Y_[l], R_[l]);
    linalg::AXPBY(c, 1.0, X_[l], -1.0, R_[l]);
  }
  P_[l - 1]->MultTranspose(R_[l], X_[l - 1]);
  if (A_[l - 1]->NumEssentialTrueDofs())
    linalg::SetSubVector(c, X_[l - 1], A_[l - 1]->GetEssentialTrueDofs(), A_[l - 1]->NumEssentialTrueDofs(), 0.0);
  VCycle(l - 1, false);
  P_[l - 1]->Mult(Y_[l - 1], R_[l]);
  linalg::AXPY(c, 1.0, R_[l], Y_[l]);
  B_[l]->SetInitialGuess(true);
  B_[l]->MultTranspose2(X_[l], Y_[l], R_[l]);
}

}  // namespace palace
