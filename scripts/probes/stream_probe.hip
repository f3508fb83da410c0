// Stand-alone probe (not part of the library): what does this GPU reach on the access patterns of the vector kernels?
//   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe && ./stream_probe
// Variants of y = a x + b y (16 B read + 8 B written per entry) and of a plain copy, interleaved rounds, median / min.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e = (x);                                                                \
    if (e != hipSuccess) {                                                             \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e));                        \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// grid-stride, one 16-B lane entry per iteration (the library's k_ew<2>)
__global__ __launch_bounds__(256) void axpby_v1(const d2 *__restrict__ x, d2 *__restrict__ y, long long n, double a, double b) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}
// U entries per thread and iteration, all loads before the first store
template <int U>
__global__ __launch_bounds__(256) void axpby_unroll(const d2 *__restrict__ x, d2 *__restrict__ y, long long n, double a, double b) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[i + u * stride], yv[u] = y[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) y[i + u * stride] = a * xv[u] + b * yv[u];
  }
  for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}
// the same with non-temporal loads / stores
template <int U>
__global__ __launch_bounds__(256) void axpby_nt(const d2 *__restrict__ x, d2 *__restrict__ y, long long n, double a, double b) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = __builtin_nontemporal_load(&x[i + u * stride]), yv[u] = __builtin_nontemporal_load(&y[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(a * xv[u] + b * yv[u], &y[i + u * stride]);
  }
  for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}
// contiguous chunk per block instead of a grid stride (each block walks its own range)
template <int U>
__global__ __launch_bounds__(256) void axpby_chunk(const d2 *__restrict__ x, d2 *__restrict__ y, long long n, double a, double b) {
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = per * blockIdx.x, hi = std::min(n, lo + per);
  long long i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    d2 xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[i + u * 256], yv[u] = y[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) y[i + u * 256] = a * xv[u] + b * yv[u];
  }
  for (; i < hi; i += 256) y[i] = a * x[i] + b * y[i];
}
template <int U>
__global__ __launch_bounds__(256) void copy_unroll(const d2 *__restrict__ x, d2 *__restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 xv[U];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) y[i + u * stride] = xv[u];
  }
  for (; i < n; i += stride) y[i] = x[i];
}
// dot-like read-only stream (two arrays)
template <int U>
__global__ __launch_bounds__(256) void dot_unroll(const d2 *__restrict__ x, const d2 *__restrict__ y, long long n, double *out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  d2 acc = {0, 0};
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 xv[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[i + u * stride], yv[u] = y[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) acc += xv[u] * yv[u];
  }
  for (; i < n; i += stride) acc += x[i] * y[i];
  if (acc.x + acc.y == 12345.678) out[0] = acc.x;
}

int main() {
  const long long n = 1ll << 25;  // d2 entries: 2 x 512 MB
  d2 *x, *y;
  double *out;
  CK(hipMalloc(&x, n * sizeof(d2)));
  CK(hipMalloc(&y, n * sizeof(d2)));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(x, 0, n * sizeof(d2)));
  CK(hipMemset(y, 0, n * sizeof(d2)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Var {
    const char *name;
    double bytes;
    void (*run)(const d2 *, d2 *, long long, double *, int);
  };
  std::vector<Var> vars = {
      {"axpby v1 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_v1, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby v1 grid 4096", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_v1, dim3(4096), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby v1 grid 8192", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_v1, dim3(8192), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby v1 grid n/256", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_v1, dim3((unsigned)(n / 256)), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby unroll2 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_unroll<2>, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby unroll4 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_unroll<4>, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby unroll4 grid 1024", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_unroll<4>, dim3(1024), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby unroll8 grid 1024", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_unroll<8>, dim3(1024), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby nt unroll4 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_nt<4>, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby nt unroll1 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_nt<1>, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby chunk4 grid 2048", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_chunk<4>, dim3(2048), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"axpby chunk4 grid 8192", 48.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(axpby_chunk<4>, dim3(8192), dim3(256), 0, 0, x, y, n, 0.5, 0.5); }},
      {"copy unroll1 grid 2048", 32.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(copy_unroll<1>, dim3(2048), dim3(256), 0, 0, x, y, n); }},
      {"copy unroll4 grid 2048", 32.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(copy_unroll<4>, dim3(2048), dim3(256), 0, 0, x, y, n); }},
      {"copy unroll1 grid n/256", 32.0, [](const d2 *x, d2 *y, long long n, double *, int) { hipLaunchKernelGGL(copy_unroll<1>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, x, y, n); }},
      {"dot unroll1 grid 2048", 32.0, [](const d2 *x, d2 *y, long long n, double *o, int) { hipLaunchKernelGGL(dot_unroll<1>, dim3(2048), dim3(256), 0, 0, x, (const d2 *)y, n, o); }},
      {"dot unroll4 grid 2048", 32.0, [](const d2 *x, d2 *y, long long n, double *o, int) { hipLaunchKernelGGL(dot_unroll<4>, dim3(2048), dim3(256), 0, 0, x, (const d2 *)y, n, o); }},
      {"hipMemcpyDtoD", 32.0, [](const d2 *x, d2 *y, long long n, double *, int) { (void)hipMemcpyAsync(y, x, n * sizeof(d2), hipMemcpyDeviceToDevice, 0); }},
  };
  // small-vector regime as well (1.25M dofs: the per-GPU share of the 8-GPU case)
  const long long sizes[2] = {n, 625000};
  for (int si = 0; si < 2; si++) {
    const long long nn = sizes[si];
    std::vector<std::vector<float>> ms(vars.size());
    const int rounds = 7, reps = si == 0 ? 4 : 40;
    for (int r = 0; r < rounds; r++)
      for (size_t v = 0; v < vars.size(); v++) {
        vars[v].run(x, y, nn, out, 0);
        CK(hipEventRecord(e0, 0));
        for (int k = 0; k < reps; k++) vars[v].run(x, y, nn, out, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms[v].push_back(t / reps);
      }
    std::printf("---- %lld entries of 16 B per array\n", nn);
    for (size_t v = 0; v < vars.size(); v++) {
      std::sort(ms[v].begin(), ms[v].end());
      const double med = ms[v][ms[v].size() / 2], mn = ms[v][0];
      std::printf("%-28s median %8.4f ms  %7.1f GB/s   best %7.1f GB/s\n", vars[v].name, med, vars[v].bytes * nn / med / 1e6,
                  vars[v].bytes * nn / mn / 1e6);
    }
  }
  return 0;
}
