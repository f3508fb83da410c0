#include "complex.hpp"

#include <algorithm>

#include <cstdlib>
#include <string>

#include <cmath>
#include <cstdio>

#include "comm.hpp"
#include "krylov_impl.hpp"

namespace palace {

namespace {

constexpr int kB = 256, kMaxB = 1024;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// partial sums of (xr.yr + xi.yi, xi.yr - xr.yi): both parts of y^H x in one pass
__global__ void k_cdot_partial(const double *__restrict__ xr, const double *__restrict__ xi,
                               const double *__restrict__ yr, const double *__restrict__ yi, long long n,
                               double *__restrict__ partial) {
  __shared__ double pr[kB / 64], pi[kB / 64];
  double sr = 0.0, si = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = xr[i], b = xi[i], c = yr[i], d = yi[i];
    sr += a * c + b * d;
    si += b * c - a * d;
  }
  sr = wsum(sr), si = wsum(si);
  if ((threadIdx.x & 63) == 0) pr[threadIdx.x >> 6] = sr, pi[threadIdx.x >> 6] = si;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kB / 64; i++) a += pr[i], b += pi[i];
    partial[2 * blockIdx.x] = a, partial[2 * blockIdx.x + 1] = b;
  }
}
__global__ void k_cdot_final(const double *__restrict__ partial, int nb, double *__restrict__ out) {
  __shared__ double pr[kB / 64], pi[kB / 64];
  double sr = 0.0, si = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) sr += partial[2 * i], si += partial[2 * i + 1];
  sr = wsum(sr), si = wsum(si);
  if ((threadIdx.x & 63) == 0) pr[threadIdx.x >> 6] = sr, pi[threadIdx.x >> 6] = si;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kB / 64; i++) a += pr[i], b += pi[i];
    out[0] = a, out[1] = b;
  }
}
// the same for x^T y: (xr.yr - xi.yi, xi.yr + xr.yi)
__global__ void k_ctdot_partial(const double *__restrict__ xr, const double *__restrict__ xi,
                                const double *__restrict__ yr, const double *__restrict__ yi, long long n,
                                double *__restrict__ partial) {
  __shared__ double pr[kB / 64], pi[kB / 64];
  double sr = 0.0, si = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = xr[i], b = xi[i], c = yr[i], d = yi[i];
    sr += a * c - b * d;
    si += b * c + a * d;
  }
  sr = wsum(sr), si = wsum(si);
  if ((threadIdx.x & 63) == 0) pr[threadIdx.x >> 6] = sr, pi[threadIdx.x >> 6] = si;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < kB / 64; i++) a += pr[i], b += pi[i];
    partial[2 * blockIdx.x] = a, partial[2 * blockIdx.x + 1] = b;
  }
}
// z = a x + b y + c z with complex coefficients; HAS_Y = false: z = a x + c z
template <bool HAS_Y>
__global__ void k_caxpbypcz(double ar, double ai, const double *__restrict__ xr, const double *__restrict__ xi, double br,
                            double bi, const double *__restrict__ yr, const double *__restrict__ yi, double cr, double ci,
                            double *__restrict__ zr, double *__restrict__ zi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double x0 = xr[i], x1 = xi[i], z0 = zr[i], z1 = zi[i];
    double r = ar * x0 - ai * x1 + cr * z0 - ci * z1, m = ai * x0 + ar * x1 + ci * z0 + cr * z1;
    if (HAS_Y) {
      const double y0 = yr[i], y1 = yi[i];
      r += br * y0 - bi * y1, m += bi * y0 + br * y1;
    }
    zr[i] = r, zi[i] = m;
  }
}
// x = |x| ; x = 1 ./ x ; x = s y (complex s) ; y (+)= a d .* x with d optionally conjugated
__global__ void k_cabs(double *__restrict__ xr, double *__restrict__ xi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    xr[i] = sqrt(xr[i] * xr[i] + xi[i] * xi[i]), xi[i] = 0.0;
}
__global__ void k_crecip(double *__restrict__ xr, double *__restrict__ xi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = xr[i], b = xi[i], s = 1.0 / (a * a + b * b);
    xr[i] = a * s, xi[i] = -b * s;
  }
}
__global__ void k_cset_scaled(double sr, double si, const double *__restrict__ yr, const double *__restrict__ yi,
                              double *__restrict__ xr, double *__restrict__ xi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = yr[i], b = yi[i];
    xr[i] = sr * a - si * b, xi[i] = si * a + sr * b;
  }
}
__global__ void k_cdiag(double ar, double ai, const double *__restrict__ dr, const double *__restrict__ di, double sgn,
                        const double *__restrict__ xr, const double *__restrict__ xi, double *__restrict__ yr,
                        double *__restrict__ yi, int add, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double p = dr[i], q = sgn * di[i], a = xr[i], b = xi[i];
    const double tr = p * a - q * b, ti = q * a + p * b;  // d .* x
    const double r = ar * tr - ai * ti, m = ai * tr + ar * ti;
    yr[i] = add ? yr[i] + r : r, yi[i] = add ? yi[i] + m : m;
  }
}
// d = sd d + sr dinv .* r (chebyshev.cpp:69-156)
__global__ void k_ccheb(double sd, double sr, const double *__restrict__ pr, const double *__restrict__ pi,
                        const double *__restrict__ rr, const double *__restrict__ ri, double *__restrict__ dr,
                        double *__restrict__ di, int first, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double p = pr[i], q = pi[i], a = rr[i], b = ri[i];
    const double tr = sr * (p * a - q * b), ti = sr * (q * a + p * b);
    dr[i] = first ? tr : sd * dr[i] + tr, di[i] = first ? ti : sd * di[i] + ti;
  }
}
// y += alpha x (complex alpha)
__global__ void k_caxpy(double ar, double ai, const double *__restrict__ xr, const double *__restrict__ xi,
                        double *__restrict__ yr, double *__restrict__ yi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double a = xr[i], b = xi[i];
    yr[i] += ar * a - ai * b;
    yi[i] += ai * a + ar * b;
  }
}
__global__ void k_cscale(double s, double *__restrict__ xr, double *__restrict__ xi, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    xr[i] *= s, xi[i] *= s;
}

struct CScratch {
  double *d, *h;
};
CScratch cscratch(const Context &c) {  // the context's reduction scratch: [2 kMaxB] partial sums, then the results
  Workspace &w = c.Work();
  return {w.Device(2 * kMaxB + 8), w.Pinned(4)};
}
inline int grid(long long n) { return (int)std::max(1LL, std::min<long long>((n + kB - 1) / kB, kMaxB)); }

}  // namespace

namespace linalg {

std::complex<double> Dot(const Context &c, const ComplexVector &x, const ComplexVector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in complex Dot");
  StreamGraph::RequireNotRecording("linalg::Dot");
  const CScratch s = cscratch(c);
  const int nb = grid(x.Size());
  hipLaunchKernelGGL(k_cdot_partial, dim3(nb), dim3(kB), 0, c.stream, x.Real().Data(), x.Imag().Data(),
                     y.Real().Data(), y.Imag().Data(), (long long)x.Size(), s.d);
  hipLaunchKernelGGL(k_cdot_final, dim3(1), dim3(kB), 0, c.stream, s.d, nb, s.d + 2 * kMaxB);
  PA_HIP(hipGetLastError());
  if (c.comm) c.comm->AllReduceSum(s.d + 2 * kMaxB, 2, c.stream);
  PA_HIP(hipMemcpyAsync(s.h, s.d + 2 * kMaxB, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
  if (c.comm) c.comm->PeerCheckNow();
  return {s.h[0], s.h[1]};
}
double Norml2(const Context &c, const ComplexVector &x) { return std::sqrt(std::abs(Dot(c, x, x).real())); }
void AXPY(const Context &c, std::complex<double> a, const ComplexVector &x, ComplexVector &y) {
  hipLaunchKernelGGL(k_caxpy, dim3(grid(x.Size())), dim3(kB), 0, c.stream, a.real(), a.imag(), x.Real().Data(),
                     x.Imag().Data(), y.Real().Data(), y.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void Scale(const Context &c, double s, ComplexVector &x) {
  hipLaunchKernelGGL(k_cscale, dim3(grid(x.Size())), dim3(kB), 0, c.stream, s, x.Real().Data(), x.Imag().Data(),
                     (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void Copy(const Context &c, const ComplexVector &x, ComplexVector &y) {
  Copy(c, x.Real(), y.Real());
  Copy(c, x.Imag(), y.Imag());
}
void Fill(const Context &c, ComplexVector &x, double s) {
  Fill(c, x.Real(), s);
  Fill(c, x.Imag(), s);
}
void SetSubVector(const Context &c, ComplexVector &x, const int32_t *d_rows, int nrows, double s) {
  SetSubVector(c, x.Real(), d_rows, nrows, s);
  SetSubVector(c, x.Imag(), d_rows, nrows, s);
}
void SetSubVector(const Context &c, ComplexVector &x, const int32_t *d_rows, int nrows, const ComplexVector &y) {
  SetSubVector(c, x.Real(), d_rows, nrows, y.Real());
  SetSubVector(c, x.Imag(), d_rows, nrows, y.Imag());
}
void Conj(const Context &c, ComplexVector &x) { Scale(c, -1.0, x.Imag()); }
std::complex<double> TransposeDot(const Context &c, const ComplexVector &x, const ComplexVector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in complex TransposeDot");
  StreamGraph::RequireNotRecording("linalg::TransposeDot");
  const CScratch s = cscratch(c);
  const int nb = grid(x.Size());
  hipLaunchKernelGGL(k_ctdot_partial, dim3(nb), dim3(kB), 0, c.stream, x.Real().Data(), x.Imag().Data(),
                     y.Real().Data(), y.Imag().Data(), (long long)x.Size(), s.d);
  hipLaunchKernelGGL(k_cdot_final, dim3(1), dim3(kB), 0, c.stream, s.d, nb, s.d + 2 * kMaxB);
  PA_HIP(hipGetLastError());
  if (c.comm) c.comm->AllReduceSum(s.d + 2 * kMaxB, 2, c.stream);
  PA_HIP(hipMemcpyAsync(s.h, s.d + 2 * kMaxB, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
  if (c.comm) c.comm->PeerCheckNow();
  return {s.h[0], s.h[1]};
}
void Scale(const Context &c, std::complex<double> s, ComplexVector &x) {
  if (s.imag() == 0.0) return Scale(c, s.real(), x);
  hipLaunchKernelGGL(k_cset_scaled, dim3(grid(x.Size())), dim3(kB), 0, c.stream, s.real(), s.imag(), x.Real().Data(),
                     x.Imag().Data(), x.Real().Data(), x.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void Abs(const Context &c, ComplexVector &x) {
  hipLaunchKernelGGL(k_cabs, dim3(grid(x.Size())), dim3(kB), 0, c.stream, x.Real().Data(), x.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void Reciprocal(const Context &c, ComplexVector &x) {
  hipLaunchKernelGGL(k_crecip, dim3(grid(x.Size())), dim3(kB), 0, c.stream, x.Real().Data(), x.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void AXPBY(const Context &c, std::complex<double> alpha, const ComplexVector &x, std::complex<double> beta, ComplexVector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in complex AXPBY");
  hipLaunchKernelGGL(k_caxpbypcz<false>, dim3(grid(x.Size())), dim3(kB), 0, c.stream, alpha.real(), alpha.imag(),
                     x.Real().Data(), x.Imag().Data(), 0.0, 0.0, (const double *)nullptr, (const double *)nullptr, beta.real(),
                     beta.imag(), y.Real().Data(), y.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void AXPBYPCZ(const Context &c, std::complex<double> alpha, const ComplexVector &x, std::complex<double> beta,
              const ComplexVector &y, std::complex<double> gamma, ComplexVector &z) {
  PA_REQUIRE(x.Size() == y.Size() && x.Size() == z.Size(), "size mismatch in complex AXPBYPCZ");
  hipLaunchKernelGGL(k_caxpbypcz<true>, dim3(grid(x.Size())), dim3(kB), 0, c.stream, alpha.real(), alpha.imag(),
                     x.Real().Data(), x.Imag().Data(), beta.real(), beta.imag(), y.Real().Data(), y.Imag().Data(), gamma.real(),
                     gamma.imag(), z.Real().Data(), z.Imag().Data(), (long long)x.Size());
  PA_HIP(hipGetLastError());
}
void SetBlocks(const Context &c, ComplexVector &x, const std::vector<const ComplexVector *> &y,
               const std::vector<std::complex<double>> &s) {
  PA_REQUIRE(s.empty() || y.size() == s.size(), "Mismatch in dimension of vector blocks and scaling coefficients!");
  long long off = 0;
  for (size_t b = 0; b < y.size(); b++) {
    PA_REQUIRE(y[b] && off + y[b]->Size() <= x.Size() && (b + 1 < y.size() || off + y[b]->Size() == x.Size()),
               "Mismatch between sum of block dimensions and parent vector dimension!");
    const std::complex<double> sb = s.empty() ? 1.0 : s[b];
    const long long n = y[b]->Size();
    if (n)
      hipLaunchKernelGGL(k_cset_scaled, dim3(grid(n)), dim3(kB), 0, c.stream, sb.real(), sb.imag(), y[b]->Real().Data(),
                         y[b]->Imag().Data(), x.Real().Data() + off, x.Imag().Data() + off, n);
    off += n;
  }
  PA_HIP(hipGetLastError());
}
void OrthogonalizeColumn(const Context &c, Orthogonalization kind, const std::vector<ComplexVector> &V, ComplexVector &w,
                         std::complex<double> *H, int m, const Operator *weight) {
  PA_REQUIRE(m >= 0 && (size_t)m <= V.size(), "Out of bounds number of columns for orthogonalization!");
  if (m == 0) return;
  for (int j = 0; j < m; j++) PA_REQUIRE(V[j].Size() == w.Size(), "size mismatch in OrthogonalizeColumn");
  ComplexVector ws;
  if (weight) {
    PA_REQUIRE(weight->Height() == w.Size() && weight->Width() == w.Size(), "weight operator does not match the vectors");
    ws.SetSize(w.Size());
  }
  auto weighted = [&]() -> const ComplexVector & {
    if (!weight) return w;
    weight->Mult(w.Real(), ws.Real());
    weight->Mult(w.Imag(), ws.Imag());
    return ws;
  };
  if (DeviceOrthogonalization() && !(weight && kind == Orthogonalization::MGS)) {  // orthog.hip: coefficients stay on the device
    const int passes = (weight && kind == Orthogonalization::CGS2) ? 2 : 1;
    const Orthogonalization k1 = weight ? Orthogonalization::CGS : kind;
    std::vector<std::complex<double>> dH;
    for (int pass = 0; pass < passes; pass++) {
      const ComplexVector &x = weighted();
      if (pass) dH.resize((size_t)m);
      OrthogonalizeColumnDevice(c, k1, V, w, weight ? &x : nullptr, pass ? dH.data() : H, m, false, nullptr);
    }
    for (size_t j = 0; j < dH.size(); j++) H[j] += dH[j];
    return;
  }
  if (kind == Orthogonalization::MGS) {
    for (int j = 0; j < m; j++) {
      H[j] = Dot(c, weighted(), V[j]);
      AXPY(c, -H[j], V[j], w);
    }
    return;
  }
  auto pass = [&](std::complex<double> *h) {
    const ComplexVector &x = weighted();
    for (int j = 0; j < m; j++) h[j] = Dot(c, x, V[j]);  // all from the same w
    for (int j = 0; j < m; j++) AXPY(c, -h[j], V[j], w);
  };
  pass(H);
  if (kind == Orthogonalization::CGS2) {
    std::vector<std::complex<double>> dH((size_t)m);
    pass(dH.data());
    for (int j = 0; j < m; j++) H[j] += dH[j];
  }
}

}  // namespace linalg

// ---- ComplexOperator defaults (operator.cpp:17-56) -----------------------------------------------------------------
void ComplexOperator::AssembleDiagonal(ComplexVector &) const { throw pa::Error("Base class ComplexOperator does not implement AssembleDiagonal!"); }
void ComplexOperator::MultTranspose(const ComplexVector &, ComplexVector &) const { throw pa::Error("Base class ComplexOperator does not implement MultTranspose!"); }
void ComplexOperator::MultHermitianTranspose(const ComplexVector &, ComplexVector &) const { throw pa::Error("Base class ComplexOperator does not implement MultHermitianTranspose!"); }
void ComplexOperator::AddMult(const ComplexVector &, ComplexVector &, std::complex<double>) const { throw pa::Error("Base class ComplexOperator does not implement AddMult!"); }
void ComplexOperator::AddMultTranspose(const ComplexVector &, ComplexVector &, std::complex<double>) const { throw pa::Error("Base class ComplexOperator does not implement AddMultTranspose!"); }
void ComplexOperator::AddMultHermitianTranspose(const ComplexVector &, ComplexVector &, std::complex<double>) const { throw pa::Error("Base class ComplexOperator does not implement AddMultHermitianTranspose!"); }

// ---- ComplexWrapperOperator (operator.cpp:58-413) ------------------------------------------------------------------
void ComplexDiagonalOperator::Apply(const ComplexVector &x, ComplexVector &y, std::complex<double> a, bool add, bool conj) const {
  PA_REQUIRE(x.Size() == d_.Size() && y.Size() == d_.Size(), "size mismatch in ComplexDiagonalOperator");
  hipLaunchKernelGGL(k_cdiag, dim3(grid(x.Size())), dim3(kB), 0, ctx_->stream, a.real(), a.imag(), d_.Real().Data(),
                     d_.Imag().Data(), conj ? -1.0 : 1.0, x.Real().Data(), x.Imag().Data(), y.Real().Data(), y.Imag().Data(),
                     add ? 1 : 0, (long long)x.Size());
  PA_HIP(hipGetLastError());
}

ComplexWrapperOperator::ComplexWrapperOperator(const Context &ctx, const Operator *Ar, const Operator *Ai)
    : ctx_(&ctx), Ar_(Ar), Ai_(Ai) {
  PA_REQUIRE(Ar || Ai, "Cannot construct ComplexWrapperOperator from an empty matrix!");
  PA_REQUIRE(!Ar || !Ai || (Ar->Height() == Ai->Height() && Ar->Width() == Ai->Width()),
             "Mismatch in dimension of real and imaginary matrix parts!");
  height = Ar ? Ar->Height() : Ai->Height();
  width = Ar ? Ar->Width() : Ai->Width();
  t_.SetSize(height);
  const auto *cr = dynamic_cast<const ceed::Operator *>(Ar), *ci = dynamic_cast<const ceed::Operator *>(Ai);
  fused_ = cr && ci && ceed::Operator::ComplexFused(*cr, *ci) != 0;
  const auto *pr = dynamic_cast<const ParOperator *>(Ar), *pi = dynamic_cast<const ParOperator *>(Ai);
  if (pr && pi && !pr->GetHalo() && !pi->GetHalo()) {
    const auto *lr = dynamic_cast<const ceed::Operator *>(&pr->LocalOperator());
    const auto *li = dynamic_cast<const ceed::Operator *>(&pi->LocalOperator());
    const int ne = pr->NumEssentialTrueDofs();
    const int kind = (lr && li) ? ceed::Operator::ComplexFused(*lr, *li) : 0;
    bool same = kind != 0 && ne == pi->NumEssentialTrueDofs() && (ne == 0 || (pr->FusesEssential() && (kind == 1 || pi->FusesEssential()))) &&  // (round 5: the forms with further sub-operators too)
                (ne == 0 || pi->GetDiagonalPolicy() == ParOperator::DiagonalPolicy::DIAG_ZERO);
    if (same && ne) {  // the two lists, once
      std::vector<int32_t> a((size_t)ne), b((size_t)ne);
      PA_HIP(hipMemcpy(a.data(), pr->GetEssentialTrueDofs(), sizeof(int32_t) * ne, hipMemcpyDeviceToHost));
      PA_HIP(hipMemcpy(b.data(), pi->GetEssentialTrueDofs(), sizeof(int32_t) * ne, hipMemcpyDeviceToHost));
      std::sort(a.begin(), a.end()), std::sort(b.begin(), b.end());
      same = a == b;
    }
    if (same) {
      par_fused_r_ = lr, par_fused_i_ = li;
      par_fused_policy_ = ne ? (pr->GetDiagonalPolicy() == ParOperator::DiagonalPolicy::DIAG_ONE ? 1 : 0) : -1;
    }
  }
}

void ComplexWrapperOperator::AssembleDiagonal(ComplexVector &diag) const {
  linalg::Fill(*ctx_, diag, 0.0);
  if (Ar_) Ar_->AssembleDiagonal(diag.Real());
  if (Ai_) Ai_->AssembleDiagonal(diag.Imag());
}

void ComplexWrapperOperator::AddReal(const Operator *A, bool transpose, const Vector &x, Vector &y, double s) const {
  // the reference calls Operator::AddMult(x, y, s); ceed::Operator only accepts s = 1 (operator.cpp:194), so go through
  // a temporary whenever the coefficient is not 1
  if (t_.Size() != y.Size()) t_.SetSize(y.Size());
  if (transpose)
    A->MultTranspose(x, t_);
  else
    A->Mult(x, t_);
  linalg::AXPY(*ctx_, s, t_, y);
}

void ComplexWrapperOperator::Mult(const ComplexVector &x, ComplexVector &y) const {
  // operator.cpp:98-134: yr = Ar xr - Ai xi, yi = Ai xr + Ar xi
  // Each real operator meets both parts of x: with ParOperators the pair goes through one pass over the
  // element data (ParOperator::Mult2), otherwise through two applies as in the reference.
  const Context &c = *ctx_;
  if (par_fused_r_ && x.Real().Data() != y.Real().Data()) {
    ceed::Operator::MultComplex(*par_fused_r_, *par_fused_i_, x.Real(), x.Imag(), y.Real(), y.Imag(), par_fused_policy_);
    return;
  }
  if (fused_) {  // both parts in one pass over the element data
    ceed::Operator::MultComplex(*static_cast<const ceed::Operator *>(Ar_), *static_cast<const ceed::Operator *>(Ai_), x.Real(),
                                x.Imag(), y.Real(), y.Imag());
    return;
  }
  // whether a pair of applies shares one pass over the element data is ParOperator::Mult2's decision (it does when the
  // operator has no streaming form); PALACE_AMD_MULT2=0 forces separate applies for A/B runs
  static const bool pair = !(getenv("PALACE_AMD_MULT2") && std::string(getenv("PALACE_AMD_MULT2")) == "0");
  const auto *par_i = pair ? dynamic_cast<const ParOperator *>(Ai_) : nullptr;
  const auto *par_r = pair ? dynamic_cast<const ParOperator *>(Ar_) : nullptr;
  if (Ar_) {  // yr = Ar xr, yi = Ar xi
    if (par_r) {
      par_r->Mult2(x.Real(), x.Imag(), y.Real(), y.Imag());
    } else {
      Ar_->Mult(x.Real(), y.Real());
      Ar_->Mult(x.Imag(), y.Imag());
    }
  } else {
    linalg::Fill(c, y, 0.0);
  }
  if (Ai_) {  // yr -= Ai xi, yi += Ai xr
    if (t_.Size() != height) t_.SetSize(height);
    if (t2_.Size() != height) t2_.SetSize(height);
    if (par_i) {
      par_i->Mult2(x.Imag(), x.Real(), t_, t2_);
    } else {
      Ai_->Mult(x.Imag(), t_);
      Ai_->Mult(x.Real(), t2_);
    }
    linalg::AXPY(c, -1.0, t_, y.Real());
    linalg::AXPY(c, 1.0, t2_, y.Imag());
  }
}

void ComplexWrapperOperator::MultTranspose(const ComplexVector &x, ComplexVector &y) const {
  // operator.cpp:136-176: yr = Ar^T xr - Ai^T xi, yi = Ai^T xr + Ar^T xi
  const Context &c = *ctx_;
  if (Ai_) {
    Ai_->MultTranspose(x.Imag(), y.Real());
    linalg::AXPBY(c, 0.0, y.Real(), -1.0, y.Real());
    Ai_->MultTranspose(x.Real(), y.Imag());
  } else {
    linalg::Fill(c, y, 0.0);
  }
  if (Ar_) {
    AddReal(Ar_, true, x.Real(), y.Real(), 1.0);
    AddReal(Ar_, true, x.Imag(), y.Imag(), 1.0);
  }
}

void ComplexWrapperOperator::MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const {
  // operator.cpp:178-219: yr = Ar^T xr + Ai^T xi, yi = -Ai^T xr + Ar^T xi
  const Context &c = *ctx_;
  if (Ai_) {
    Ai_->MultTranspose(x.Imag(), y.Real());
    Ai_->MultTranspose(x.Real(), y.Imag());
    linalg::AXPBY(c, 0.0, y.Imag(), -1.0, y.Imag());
  } else {
    linalg::Fill(c, y, 0.0);
  }
  if (Ar_) {
    AddReal(Ar_, true, x.Real(), y.Real(), 1.0);
    AddReal(Ar_, true, x.Imag(), y.Imag(), 1.0);
  }
}

void ComplexWrapperOperator::AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // operator.cpp:221-283
  if (a.real() != 0.0 && a.imag() != 0.0) {
    if (ty_.Size() != height) ty_.SetSize(height);
    Mult(x, ty_);
    linalg::AXPY(*ctx_, a, ty_, y);
  } else if (a.real() != 0.0) {
    if (Ar_) AddReal(Ar_, false, x.Real(), y.Real(), a.real()), AddReal(Ar_, false, x.Imag(), y.Imag(), a.real());
    if (Ai_) AddReal(Ai_, false, x.Imag(), y.Real(), -a.real()), AddReal(Ai_, false, x.Real(), y.Imag(), a.real());
  } else if (a.imag() != 0.0) {
    if (Ar_) AddReal(Ar_, false, x.Real(), y.Imag(), a.imag()), AddReal(Ar_, false, x.Imag(), y.Real(), -a.imag());
    if (Ai_) AddReal(Ai_, false, x.Imag(), y.Imag(), -a.imag()), AddReal(Ai_, false, x.Real(), y.Real(), -a.imag());
  }
}

void ComplexWrapperOperator::AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // operator.cpp:285-347
  if (a.real() != 0.0 && a.imag() != 0.0) {
    if (tx_.Size() != width) tx_.SetSize(width);
    MultTranspose(x, tx_);
    linalg::AXPY(*ctx_, a, tx_, y);
  } else if (a.real() != 0.0) {
    if (Ar_) AddReal(Ar_, true, x.Real(), y.Real(), a.real()), AddReal(Ar_, true, x.Imag(), y.Imag(), a.real());
    if (Ai_) AddReal(Ai_, true, x.Imag(), y.Real(), -a.real()), AddReal(Ai_, true, x.Real(), y.Imag(), a.real());
  } else if (a.imag() != 0.0) {
    if (Ar_) AddReal(Ar_, true, x.Real(), y.Imag(), a.imag()), AddReal(Ar_, true, x.Imag(), y.Real(), -a.imag());
    if (Ai_) AddReal(Ai_, true, x.Imag(), y.Imag(), -a.imag()), AddReal(Ai_, true, x.Real(), y.Real(), -a.imag());
  }
}

void ComplexWrapperOperator::AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // operator.cpp:349-413
  if (a.real() != 0.0 && a.imag() != 0.0) {
    if (tx_.Size() != width) tx_.SetSize(width);
    MultHermitianTranspose(x, tx_);
    linalg::AXPY(*ctx_, a, tx_, y);
  } else if (a.real() != 0.0) {
    if (Ar_) AddReal(Ar_, true, x.Real(), y.Real(), a.real()), AddReal(Ar_, true, x.Imag(), y.Imag(), a.real());
    if (Ai_) AddReal(Ai_, true, x.Imag(), y.Real(), a.real()), AddReal(Ai_, true, x.Real(), y.Imag(), -a.real());
  } else if (a.imag() != 0.0) {
    if (Ar_) AddReal(Ar_, true, x.Real(), y.Imag(), a.imag()), AddReal(Ar_, true, x.Imag(), y.Real(), -a.imag());
    if (Ai_) AddReal(Ai_, true, x.Imag(), y.Imag(), a.imag()), AddReal(Ai_, true, x.Real(), y.Real(), a.imag());
  }
}

// ---- ComplexParOperator (rap.cpp:393-749) --------------------------------------------------------------------------
ComplexParOperator::ComplexParOperator(const Context &ctx, const Operator *Ar, const Operator *Ai, int n_true, const Halo *halo)
    : ComplexOperator(n_true, n_true), ctx_(&ctx), Ar_(Ar), Ai_(Ai), halo_(halo), n_true_(n_true) {
  A_ = std::make_unique<ComplexWrapperOperator>(ctx, Ar, Ai);
  PA_REQUIRE(A_->Height() == A_->Width(), "ComplexParOperator needs square local operators");
  n_local_ = A_->Height();
  PA_REQUIRE(n_true <= n_local_ && (halo || n_true == n_local_), "local != true dofs requires a halo plan");
  if (Ar) RAPr_ = std::make_unique<ParOperator>(ctx, *Ar, n_true, nullptr, 0, ParOperator::DiagonalPolicy::DIAG_ONE, halo);
  if (Ai) RAPi_ = std::make_unique<ParOperator>(ctx, *Ai, n_true, nullptr, 0, ParOperator::DiagonalPolicy::DIAG_ZERO, halo);
  RAP_ = std::make_unique<ComplexWrapperOperator>(ctx, RAPr_.get(), RAPi_.get());
  lx_.SetSize(n_local_), ly_.SetSize(n_local_);
  UpdateFused();
}
void ComplexParOperator::UpdateFused() {
  fused_r_ = fused_i_ = nullptr;
  const auto *cr = dynamic_cast<const ceed::Operator *>(Ar_), *ci = dynamic_cast<const ceed::Operator *>(Ai_);
  const int kind = (!halo_ && cr && ci) ? ceed::Operator::ComplexFused(*cr, *ci) : 0;
  if (!kind) return;
  // essential dofs inside the kernel (hexahedral form; round 5: dense blocks too, through the flagged index copy of Ar's block and
  // its gathers): only if this wrapper's list is the one fused into Ar
  if (n_ess_ && !(RAPr_ && RAPr_->FusesEssential())) return;
  // (dense form: the imaginary operator may carry further sub-operators -- surface terms -- that mask x through their own index copies)
  if (n_ess_ && kind != 1 && !(RAPi_ && RAPi_->FusesEssential())) return;
  fused_r_ = cr, fused_i_ = ci;
}
ComplexParOperator::~ComplexParOperator() {
  if (d_ess_) (void)hipFree(d_ess_);
}

void ComplexParOperator::SetEssentialTrueDofs(const int32_t *ess_host, int n_ess, ParOperator::DiagonalPolicy policy) {
  StreamGraph::Invalidate();
  PA_REQUIRE(policy != ParOperator::DiagonalPolicy::DIAG_ONE || RAPr_,
             "DiagonalPolicy::DIAG_ONE specified for ComplexParOperator with no real part!");
  for (int i = 0; i < n_ess; i++) PA_REQUIRE(ess_host[i] >= 0 && ess_host[i] < n_true_, "essential dof out of range");
  if (d_ess_) (void)hipFree(d_ess_);
  d_ess_ = n_ess ? pa::dev_upload(ess_host, (size_t)n_ess, ctx_->stream) : nullptr;
  n_ess_ = n_ess, policy_ = policy;
  // the real ParOperators are rebuilt with the list (real part: the policy; imaginary part: DIAG_ZERO, rap.cpp:450-457)
  if (Ar_) RAPr_ = std::make_unique<ParOperator>(*ctx_, *Ar_, n_true_, ess_host, n_ess, policy, halo_);
  if (Ai_) RAPi_ = std::make_unique<ParOperator>(*ctx_, *Ai_, n_true_, ess_host, n_ess, ParOperator::DiagonalPolicy::DIAG_ZERO, halo_);
  RAP_ = std::make_unique<ComplexWrapperOperator>(*ctx_, RAPr_.get(), RAPi_.get());
  UpdateFused();
}

ParOperator::DiagonalPolicy ComplexParOperator::GetDiagonalPolicy() const {
  PA_REQUIRE(n_ess_ > 0, "There is no DiagonalPolicy if no essential dofs have been set!");
  return policy_;
}

void ComplexParOperator::AssembleDiagonal(ComplexVector &diag) const {
  linalg::Fill(*ctx_, diag, 0.0);
  if (RAPr_) RAPr_->AssembleDiagonal(diag.Real());
  if (RAPi_) RAPi_->AssembleDiagonal(diag.Imag());
}

void ComplexParOperator::Prolongate(const ComplexVector &x, ComplexVector &lx) const {
  const Context &c = *ctx_;
  ComplexVector tx(lx.Real().Data(), lx.Imag().Data(), n_true_);
  linalg::Copy(c, x, tx);
  if (n_ess_) linalg::SetSubVector(c, tx, d_ess_, n_ess_, 0.0);
  if (halo_) {
    halo_->Prolongate(lx.Real().Data(), c.stream);
    halo_->Prolongate(lx.Imag().Data(), c.stream);
  }
}

void ComplexParOperator::RestrictFix(const ComplexVector &x, ComplexVector &ly, ComplexVector &y) const {
  const Context &c = *ctx_;
  if (halo_) {
    halo_->RestrictAdd(ly.Real().Data(), c.stream);
    halo_->RestrictAdd(ly.Imag().Data(), c.stream);
  }
  ComplexVector ty(ly.Real().Data(), ly.Imag().Data(), n_true_);
  linalg::Copy(c, ty, y);
  if (n_ess_) {
    if (policy_ == ParOperator::DiagonalPolicy::DIAG_ONE)
      linalg::SetSubVector(c, y, d_ess_, n_ess_, x);
    else
      linalg::SetSubVector(c, y, d_ess_, n_ess_, 0.0);
  }
}

void ComplexParOperator::Mult(const ComplexVector &x, ComplexVector &y) const {
  // rap.cpp:483-519.  One rank: the same result through the two real ParOperators (yr = RAPr xr - RAPi xi has xr on the
  // essential rows, yi = RAPi xr + RAPr xi has xi, with RAPi's rows zero), BC masking fused into the element kernels.
  if (!halo_ && x.Real().Data() != y.Real().Data()) {
    if (fused_r_) {  // essential dofs handled inside the fused kernel and its gathers
      ceed::Operator::MultComplex(*fused_r_, *fused_i_, x.Real(), x.Imag(), y.Real(), y.Imag(),
                                  n_ess_ ? (policy_ == ParOperator::DiagonalPolicy::DIAG_ONE ? 1 : 0) : -1);
      return;
    }
    // (a fused local operator without fused essential dofs -- dense blocks: copy / mask / apply / fix below)
    if (!A_->Fused()) return RAP_->Mult(x, y);
  }
  Prolongate(x, lx_);
  A_->Mult(lx_, ly_);
  RestrictFix(x, ly_, y);
}

void ComplexParOperator::MultTranspose(const ComplexVector &x, ComplexVector &y) const {
  // rap.cpp:521-556 (conforming spaces: the restriction transposed is the prolongation)
  Prolongate(x, ly_);
  A_->MultTranspose(ly_, lx_);
  RestrictFix(x, lx_, y);
}

void ComplexParOperator::MultHermitianTranspose(const ComplexVector &x, ComplexVector &y) const {
  // rap.cpp:558-593
  Prolongate(x, ly_);
  A_->MultHermitianTranspose(ly_, lx_);
  RestrictFix(x, lx_, y);
}

void ComplexParOperator::AddMult(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // rap.cpp:595-635
  if (tt_.Size() != n_true_) tt_.SetSize(n_true_);
  Mult(x, tt_);
  linalg::AXPY(*ctx_, a, tt_, y);
}
void ComplexParOperator::AddMultTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // rap.cpp:637-677
  if (tt_.Size() != n_true_) tt_.SetSize(n_true_);
  MultTranspose(x, tt_);
  linalg::AXPY(*ctx_, a, tt_, y);
}
void ComplexParOperator::AddMultHermitianTranspose(const ComplexVector &x, ComplexVector &y, std::complex<double> a) const {
  // rap.cpp:679-719
  if (tt_.Size() != n_true_) tt_.SetSize(n_true_);
  MultHermitianTranspose(x, tt_);
  linalg::AXPY(*ctx_, a, tt_, y);
}

// ---- complex smoothers -----------------------------------------------------------------------------------------------
void ComplexSolver::Mult2(const ComplexVector &, ComplexVector &, ComplexVector &) const {
  throw pa::Error("Mult2() with temporary storage vector is not implemented for base class Solver<OperType>!");
}

namespace linalg {
double SpectralNorm(const Context &c, const ComplexOperator &A, bool herm, double tol, int max_it, uint64_t seed) {
  ComplexVector u(A.Height()), v(A.Height());
  SetRandom(c, u.Real(), seed), SetRandom(c, u.Imag(), seed + 0x51ed27ull);
  auto normalize = [&](ComplexVector &w) {
    const double nrm = Norml2(c, w);
    PA_REQUIRE(nrm > 0.0, "zero vector norm in normalization");
    Scale(c, 1.0 / nrm, w);
    return nrm;
  };
  normalize(u);
  double l = 0.0, l0 = 0.0;
  int it = 0;
  while (it < max_it) {
    A.Mult(u, v);
    if (herm)
      Copy(c, v, u);
    else
      A.MultHermitianTranspose(v, u);
    l = normalize(u);
    if (it > 0 && std::abs(l - l0) / l0 < tol) break;
    l0 = l, it++;
  }
  return herm ? l : std::sqrt(l);
}
}  // namespace linalg

void ComplexJacobiSmoother::SetOperator(const ComplexOperator &op) {
  StreamGraph::Invalidate();
  height = op.Height(), width = op.Width();
  dinv_.SetSize(height);
  op.AssembleDiagonal(dinv_);
  linalg::Reciprocal(*ctx_, dinv_);
}
void ComplexJacobiSmoother::Mult(const ComplexVector &x, ComplexVector &y) const {
  PA_REQUIRE(!initial_guess, "JacobiSmoother is not implemented for iterative mode!");
  ComplexDiagonalOperator(*ctx_, dinv_).Mult(x, y);
}

void ComplexChebyshevSmoother::SetOperator(const ComplexOperator &op) {
  StreamGraph::Invalidate();
  A_ = &op, height = op.Height(), width = op.Width();
  dinv_.SetSize(height), d_.SetSize(height), r_.SetSize(height);
  op.AssembleDiagonal(dinv_);
  linalg::Reciprocal(*ctx_, dinv_);
  ComplexDiagonalOperator Dinv(*ctx_, dinv_);
  ComplexProductOperator DinvA(Dinv, op);
  lambda_max_ = sf_max_ * linalg::SpectralNorm(*ctx_, DinvA, op.IsReal());
  PA_REQUIRE(lambda_max_ > 0.0, "Encountered zero maximum eigenvalue in Chebyshev smoother!");
  if (!fourth_kind_) {  // chebyshev.cpp:244-255
    double sf_min = sf_min_;
    if (sf_min <= 0.0) sf_min = 1.69 / (std::pow(order_, 1.68) + 2.11 * order_ + 1.98);
    const double lambda_min = sf_min * lambda_max_;
    theta_ = 0.5 * (lambda_max_ + lambda_min), delta_ = 0.5 * (lambda_max_ - lambda_min);
  }
}
void ComplexChebyshevSmoother::Mult(const ComplexVector &x, ComplexVector &y) const { Mult2(x, y, r_); }
void ComplexChebyshevSmoother::Mult2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const {
  const Context &c = *ctx_;
  auto step = [&](double sd, double sr, bool first) {
    hipLaunchKernelGGL(k_ccheb, dim3(grid(height)), dim3(kB), 0, c.stream, sd, sr, dinv_.Real().Data(), dinv_.Imag().Data(),
                       r.Real().Data(), r.Imag().Data(), d_.Real().Data(), d_.Imag().Data(), first ? 1 : 0, (long long)height);
    PA_HIP(hipGetLastError());
  };
  for (int it = 0; it < pc_it_; it++) {
    if (initial_guess || it > 0) {
      A_->Mult(y, r);
      linalg::AXPBY(c, 1.0, x, -1.0, r);
    } else {
      linalg::Copy(c, x, r);
      linalg::Fill(c, y, 0.0);
    }
    double rhop = fourth_kind_ ? 0.0 : delta_ / theta_;
    step(0.0, fourth_kind_ ? 4.0 / (3.0 * lambda_max_) : 1.0 / theta_, true);
    for (int k = 1; k < order_; k++) {
      linalg::AXPY(c, 1.0, d_, y);
      A_->AddMult(d_, r, -1.0);
      if (fourth_kind_) {
        step((2.0 * k - 1.0) / (2.0 * k + 3.0), (8.0 * k + 4.0) / ((2.0 * k + 3.0) * lambda_max_), false);
      } else {
        const double rho = 1.0 / (2.0 * theta_ / delta_ - rhop);
        step(rho * rhop, 2.0 * rho / delta_, false);
        rhop = rho;
      }
    }
    linalg::AXPY(c, 1.0, d_, y);
  }
}

ComplexDistRelaxationSmoother::ComplexDistRelaxationSmoother(const Context &ctx, const Operator &G, int smooth_it,
                                                             int cheby_smooth_it, int cheby_order, double cheby_sf_max,
                                                             double cheby_sf_min, bool cheby_4th_kind)
    : ctx_(&ctx), pc_it_(smooth_it), G_(&G) {
  B_ = std::make_unique<ComplexChebyshevSmoother>(ctx, cheby_smooth_it, cheby_order, cheby_sf_max, cheby_4th_kind, cheby_sf_min);
  B_G_ = std::make_unique<ComplexChebyshevSmoother>(ctx, cheby_smooth_it, cheby_order, cheby_sf_max, cheby_4th_kind, cheby_sf_min);
}
void ComplexDistRelaxationSmoother::SetOperators(const ComplexOperator &op, const ComplexParOperator &op_G) {
  StreamGraph::Invalidate();
  PA_REQUIRE(op.Height() == G_->Height() && op.Width() == G_->Height() && op_G.Height() == G_->Width() &&
                 op_G.Width() == G_->Width(),
             "Invalid operator sizes for DistRelaxationSmoother!");
  A_ = &op, A_G_ = &op_G;
  x_G_.SetSize(op_G.Height()), y_G_.SetSize(op_G.Height()), r_G_.SetSize(op_G.Height()), t_.SetSize(op.Height());
  B_->SetOperator(op);
  B_G_->SetOperator(op_G);
  height = op.Height(), width = op.Width();
}
void ComplexDistRelaxationSmoother::Mult2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const {
  // distrelaxation.cpp:98-123
  const Context &c = *ctx_;
  for (int it = 0; it < pc_it_; it++) {
    B_->SetInitialGuess(initial_guess || it > 0);
    B_->Mult2(x, y, r);
    A_->Mult(y, r);
    linalg::AXPBY(c, 1.0, x, -1.0, r);
    G_->MultTranspose(r.Real(), x_G_.Real());
    G_->MultTranspose(r.Imag(), x_G_.Imag());
    if (A_G_->NumEssentialTrueDofs())
      linalg::SetSubVector(c, x_G_, A_G_->GetEssentialTrueDofs(), A_G_->NumEssentialTrueDofs(), 0.0);
    B_G_->SetInitialGuess(false);
    B_G_->Mult2(x_G_, y_G_, r_G_);
    G_->Mult(y_G_.Real(), r.Real());
    G_->Mult(y_G_.Imag(), r.Imag());
    linalg::AXPY(c, 1.0, r, y);
  }
}
void ComplexDistRelaxationSmoother::MultTranspose2(const ComplexVector &x, ComplexVector &y, ComplexVector &r) const {
  // distrelaxation.cpp:125-151
  const Context &c = *ctx_;
  B_->SetInitialGuess(true);
  for (int it = 0; it < pc_it_; it++) {
    if (initial_guess || it > 0) {
      A_->Mult(y, r);
      linalg::AXPBY(c, 1.0, x, -1.0, r);
      G_->MultTranspose(r.Real(), x_G_.Real());
      G_->MultTranspose(r.Imag(), x_G_.Imag());
    } else {
      linalg::Fill(c, y, 0.0);
      G_->MultTranspose(x.Real(), x_G_.Real());
      G_->MultTranspose(x.Imag(), x_G_.Imag());
    }
    if (A_G_->NumEssentialTrueDofs())
      linalg::SetSubVector(c, x_G_, A_G_->GetEssentialTrueDofs(), A_G_->NumEssentialTrueDofs(), 0.0);
    B_G_->SetInitialGuess(false);
    B_G_->MultTranspose2(x_G_, y_G_, r_G_);
    G_->Mult(y_G_.Real(), r.Real());
    G_->Mult(y_G_.Imag(), r.Imag());
    linalg::AXPY(c, 1.0, r, y);
    B_->MultTranspose2(x, y, r);
  }
}

// ---- complex geometric multigrid (gmg.cpp:16-205) --------------------------------------------------------------------------
ComplexGeometricMultigridSolver::ComplexGeometricMultigridSolver(const Context &ctx, std::unique_ptr<ComplexSolver> &&coarse_solver,
                                                                 const std::vector<const Operator *> &P, int cycle_it,
                                                                 int smooth_it, int cheby_order, double cheby_sf_max,
                                                                 double cheby_sf_min, bool cheby_4th_kind,
                                                                 const std::vector<const Operator *> *G)
    : ctx_(&ctx), pc_it_(cycle_it), P_(P), A_(P.size() + 1), B_(P.size() + 1), X_(P.size() + 1), Y_(P.size() + 1),
      R_(P.size() + 1) {
  PA_REQUIRE(!G || G->size() == B_.size(), "Invalid input for distributive relaxation smoother auxiliary space transfer operators!");
  B_[0] = std::move(coarse_solver);
  for (size_t l = 1; l < B_.size(); l++) {
    if (G)  // gmg.cpp:41-47
      B_[l] = std::make_unique<ComplexDistRelaxationSmoother>(ctx, *(*G)[l], smooth_it, 1, cheby_order, cheby_sf_max, cheby_sf_min,
                                                              cheby_4th_kind);
    else
      B_[l] = std::make_unique<ComplexChebyshevSmoother>(ctx, smooth_it, cheby_order, cheby_sf_max, cheby_4th_kind, cheby_sf_min);
  }
}

void ComplexGeometricMultigridSolver::SetOperators(const std::vector<const ComplexParOperator *> &ops,
                                                   const std::vector<const ComplexParOperator *> *aux_ops) {
  StreamGraph::Invalidate();
  PA_REQUIRE(ops.size() == A_.size(), "Invalid number of levels for operators in multigrid solver setup!");
  for (size_t l = 0; l < ops.size(); l++) {
    A_[l] = ops[l];
    PA_REQUIRE(A_[l]->Width() == A_[l]->Height(), "Invalid operator sizes for GeometricMultigridSolver!");
    if (l + 1 < ops.size()) PA_REQUIRE(A_[l]->Height() == P_[l]->Width(), "Prolongation / operator size mismatch");
    if (auto *dist = dynamic_cast<ComplexDistRelaxationSmoother *>(B_[l].get())) {
      PA_REQUIRE(aux_ops && aux_ops->size() == ops.size(),
                 "Distributive relaxation smoother relies on both primary space and auxiliary space operators!");
      dist->SetOperators(*A_[l], *(*aux_ops)[l]);
    } else
      B_[l]->SetOperator(*A_[l]);
    X_[l].SetSize(A_[l]->Height()), Y_[l].SetSize(A_[l]->Height()), R_[l].SetSize(A_[l]->Height());
  }
  height = width = ops.back()->Height();
}

void ComplexGeometricMultigridSolver::Mult(const ComplexVector &x, ComplexVector &y) const {
  const int L = (int)A_.size();
  linalg::Copy(*ctx_, x, X_[L - 1]);
  for (int it = 0; it < pc_it_; it++) VCycle(L - 1, it > 0);
  linalg::Copy(*ctx_, Y_[L - 1], y);
}

void ComplexGeometricMultigridSolver::VCycle(int l, bool initial_guess) const {
  // gmg.cpp:171-205; the real transfer operators act on the real and the imaginary part (gmg.cpp:147-168)
  const Context &c = *ctx_;
  B_[l]->SetInitialGuess(initial_guess);
  if (l == 0) {
    B_[l]->Mult(X_[l], Y_[l]);
    return;
  }
  B_[l]->Mult2(X_[l], Y_[l], R_[l]);
  A_[l]->Mult(Y_[l], R_[l]);
  linalg::AXPBY(c, 1.0, X_[l], -1.0, R_[l]);
  P_[l - 1]->MultTranspose(R_[l].Real(), X_[l - 1].Real());
  P_[l - 1]->MultTranspose(R_[l].Imag(), X_[l - 1].Imag());
  if (A_[l - 1]->NumEssentialTrueDofs())
    linalg::SetSubVector(c, X_[l - 1], A_[l - 1]->GetEssentialTrueDofs(), A_[l - 1]->NumEssentialTrueDofs(), 0.0);
  VCycle(l - 1, false);
  P_[l - 1]->Mult(Y_[l - 1].Real(), R_[l].Real());
  P_[l - 1]->Mult(Y_[l - 1].Imag(), R_[l].Imag());
  linalg::AXPY(c, 1.0, R_[l], Y_[l]);
  B_[l]->SetInitialGuess(true);
  B_[l]->MultTranspose2(X_[l], Y_[l], R_[l]);
}

// ---- GMRES / FGMRES on complex vectors: the shared implementation (krylov_impl.hpp) --------------------------------
namespace {
struct ComplexKrylovOps {
  using Vec = ComplexVector;
  using Scalar = std::complex<double>;
  const Context &c;
  const ComplexOperator *A_;
  const Solver *B_;
  const ComplexSolver *Bc_;
  int n;
  void Ensure(Vec &v) const {
    if (v.Size() != n) v.SetSize(n);
  }
  void A(const Vec &x, Vec &y) const { A_->Mult(x, y); }
  bool HasB() const { return B_ != nullptr || Bc_ != nullptr; }
  void B(const Vec &x, Vec &y) const {
    PhaseRange range("Preconditioner");  // iterative.cpp:247
    if (Bc_) return Bc_->Mult(x, y);
    B_->Mult(x.Real(), y.Real());  // gmg.cpp:147-168: the real preconditioner on both parts
    B_->Mult(x.Imag(), y.Imag());
  }
  void Copy(const Vec &x, Vec &y) const { linalg::Copy(c, x, y); }
  void Zero(Vec &x) const { linalg::Fill(c, x, 0.0); }
  void BMinus(const Vec &b, Vec &r) const {
    linalg::Scale(c, -1.0, r);
    linalg::AXPY(c, Scalar(1.0, 0.0), b, r);
  }
  void Axpy(Scalar a, const Vec &x, Vec &y) const { linalg::AXPY(c, a, x, y); }
  void Scale(double s, Vec &x) const { linalg::Scale(c, s, x); }
  double Norm(const Vec &x) const { return linalg::Norml2(c, x); }
  double Orthonormalize(Orthogonalization kind, const std::vector<Vec> &V, Vec &w, Scalar *H, int m) const {
    return linalg::OrthonormalizeColumn(c, kind, V, w, H, m);
  }
};
}  // namespace

void ComplexIterativeSolver::ApplyB(const ComplexVector &x, ComplexVector &y) const {
  PhaseRange range("Preconditioner");  // iterative.cpp:247
  if (Bc_) return Bc_->Mult(x, y);
  B_->Mult(x.Real(), y.Real());  // gmg.cpp:147-168: the real preconditioner on both parts
  B_->Mult(x.Imag(), y.Imag());
}

void ComplexCgSolver::Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess) const {
  // iterative.cpp:360-486 with ScalarType = std::complex<double>
  using cplx = std::complex<double>;
  const Context &c = *ctx_;
  PA_REQUIRE(A_, "Operator must be set for CgSolver::Mult!");
  const int n = A_->Height();
  if (r_.Size() != n) r_.SetSize(n), z_.SetSize(n), p_.SetSize(n);
  const bool haveB = B_ || Bc_;
  auto check = [](cplx dot, const char *msg) {  // CheckDot, iterative.cpp:27-32
    PA_REQUIRE(std::isfinite(dot.real()) && std::isfinite(dot.imag()) && dot.real() >= 0.0, msg);
  };
  cplx beta, beta_prev = 0.0;
  if (initial_guess) {
    A_->Mult(x, r_);
    linalg::AXPBY(c, cplx(1.0), b, cplx(-1.0), r_);
  } else {
    linalg::Copy(c, b, r_);
    linalg::Fill(c, x, 0.0);
  }
  if (haveB) ApplyB(r_, z_); else linalg::Copy(c, r_, z_);
  beta = linalg::Dot(c, z_, r_);
  check(beta, "PCG preconditioner is not positive definite: (Br, r) is not a finite non-negative number");
  double res = std::sqrt(std::abs(beta));
  if (initial_guess) {
    cplx beta_rhs;
    if (haveB) {
      ApplyB(b, p_);
      beta_rhs = linalg::Dot(c, p_, b);
    } else {
      beta_rhs = linalg::Norml2(c, b);  // (the norm, not its square: iterative.cpp:406-411)
    }
    check(beta_rhs, "PCG preconditioner is not positive definite: (Bb, b) is not a finite non-negative number");
    initial_res_ = std::sqrt(std::abs(beta_rhs));
  } else {
    initial_res_ = res;
  }
  const double eps = std::max(rel_tol_ * initial_res_, abs_tol_);
  converged_ = res < eps;
  int it = 0;
  for (; it < max_it_ && !converged_; it++) {
    if (print_ > 1) std::printf("  %3d KSP residual norm ||r||_B = %.6e\n", it, res);
    if (!it)
      linalg::Copy(c, z_, p_);
    else
      linalg::AXPBY(c, cplx(1.0), z_, beta / beta_prev, p_);
    A_->Mult(p_, z_);
    const cplx denom = linalg::Dot(c, z_, p_);
    check(denom, "PCG operator is not positive definite: (Ap, p) is not a finite non-negative number");
    const cplx alpha = beta / denom;
    linalg::AXPY(c, alpha, p_, x);
    linalg::AXPY(c, -alpha, z_, r_);
    beta_prev = beta;
    if (haveB) ApplyB(r_, z_); else linalg::Copy(c, r_, z_);
    beta = linalg::Dot(c, z_, r_);
    check(beta, "PCG preconditioner is not positive definite: (Br, r) is not a finite non-negative number");
    res = std::sqrt(std::abs(beta));
    converged_ = res < eps;
  }
  if (print_ > 0)
    std::printf("  PCG (complex) solver %s in %d iterations (res %.3e, initial %.3e)\n", converged_ ? "converged" : "did NOT converge",
                it, res, initial_res_);
  final_res_ = res, final_it_ = it;
}

void ComplexGmresSolver::Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess) const {
  PA_REQUIRE(A_, "Operator must be set for GmresSolver::Mult!");
  ComplexKrylovOps ops{*ctx_, A_, B_, Bc_, A_->Height()};
  krylov::Params p;
  p.rel_tol = rel_tol_, p.abs_tol = abs_tol_, p.max_it = max_it_, p.max_dim = max_dim_, p.print = print_;
  p.flexible = flexible_, p.initial_guess = initial_guess;
  p.pc_side = pc_side_ == PreconditionerSide::RIGHT ? krylov::PreconditionerSide::RIGHT : krylov::PreconditionerSide::LEFT;
  p.orthog = orthog_, p.name = flexible_ ? "FGMRES (complex)" : "GMRES (complex)";
  krylov::Result res;
  krylov::GmresMult(ops, p, b, x, V_, Z_, r_, res);
  converged_ = res.converged, initial_res_ = res.initial_res, final_res_ = res.final_res, final_it_ = res.final_it;
}

}  // namespace palace
