#include "complex.hpp"

#include <cstdlib>
#include <string>

#include <cmath>
#include <cstdio>

#include "comm.hpp"

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
  double *d = nullptr, *h = nullptr;
  CScratch() {
    d = pa::dev_alloc<double>(2 * kMaxB + 8);
    PA_HIP(hipHostMalloc(reinterpret_cast<void **>(&h), 4 * sizeof(double), hipHostMallocDefault));
  }
};
CScratch &cscratch() {
  static CScratch s;
  return s;
}
inline int grid(long long n) { return (int)std::max(1LL, std::min<long long>((n + kB - 1) / kB, kMaxB)); }

}  // namespace

namespace linalg {

std::complex<double> Dot(const Context &c, const ComplexVector &x, const ComplexVector &y) {
  PA_REQUIRE(x.Size() == y.Size(), "size mismatch in complex Dot");
  CScratch &s = cscratch();
  const int nb = grid(x.Size());
  hipLaunchKernelGGL(k_cdot_partial, dim3(nb), dim3(kB), 0, c.stream, x.Real().Data(), x.Imag().Data(),
                     y.Real().Data(), y.Imag().Data(), (long long)x.Size(), s.d);
  hipLaunchKernelGGL(k_cdot_final, dim3(1), dim3(kB), 0, c.stream, s.d, nb, s.d + 2 * kMaxB);
  PA_HIP(hipGetLastError());
  if (c.comm) c.comm->AllReduceSum(s.d + 2 * kMaxB, 2, c.stream);
  PA_HIP(hipMemcpyAsync(s.h, s.d + 2 * kMaxB, 2 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  PA_HIP(hipStreamSynchronize(c.stream));
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

ComplexWrapperOperator::ComplexWrapperOperator(const Context &ctx, const Operator *Ar, const Operator *Ai)
    : ctx_(&ctx), Ar_(Ar), Ai_(Ai) {
  PA_REQUIRE(Ar || Ai, "Cannot construct ComplexWrapperOperator from an empty matrix!");
  PA_REQUIRE(!Ar || !Ai || (Ar->Height() == Ai->Height() && Ar->Width() == Ai->Width()),
             "Mismatch in dimension of real and imaginary matrix parts!");
  height = Ar ? Ar->Height() : Ai->Height();
  width = Ar ? Ar->Width() : Ai->Width();
  t_.SetSize(height);
}

void ComplexWrapperOperator::Mult(const ComplexVector &x, ComplexVector &y) const {
  // linalg/operator.cpp:98-134: yr = Ar xr - Ai xi, yi = Ai xr + Ar xi
  // Each real operator meets both parts of x: with ParOperators the pair goes through one pass over the
  // element data (ParOperator::Mult2), otherwise through two applies as in the reference.
  const Context &c = *ctx_;
  static const bool pair = !(getenv("PALACE_AMD_MULT2") && std::string(getenv("PALACE_AMD_MULT2")) == "0");  // A/B switch
  const auto *par_i = pair ? dynamic_cast<const ParOperator *>(Ai_) : nullptr;
  const auto *par_r = pair ? dynamic_cast<const ParOperator *>(Ar_) : nullptr;
  if (Ai_) {
    if (par_i) {
      par_i->Mult2(x.Imag(), x.Real(), y.Real(), y.Imag());
    } else {
      Ai_->Mult(x.Imag(), y.Real());
      Ai_->Mult(x.Real(), y.Imag());
    }
    linalg::AXPBY(c, 0.0, y.Real(), -1.0, y.Real());
  } else {
    linalg::Fill(c, y, 0.0);
  }
  if (Ar_) {
    if (par_r) {
      if (t2_.Size() != t_.Size()) t2_.SetSize(t_.Size());
      par_r->Mult2(x.Real(), x.Imag(), t_, t2_);
      linalg::AXPY(c, 1.0, t_, y.Real());
      linalg::AXPY(c, 1.0, t2_, y.Imag());
    } else {
      Ar_->Mult(x.Real(), t_);
      linalg::AXPY(c, 1.0, t_, y.Real());
      Ar_->Mult(x.Imag(), t_);
      linalg::AXPY(c, 1.0, t_, y.Imag());
    }
  }
}

void ComplexGmresSolver::ApplyB(const ComplexVector &x, ComplexVector &y) const {
  if (B_) {
    B_->Mult(x.Real(), y.Real());
    B_->Mult(x.Imag(), y.Imag());
  } else {
    linalg::Copy(*ctx_, x, y);
  }
}

void ComplexGmresSolver::Mult(const ComplexVector &b, ComplexVector &x, bool initial_guess) const {
  using cd = std::complex<double>;
  const Context &c = *ctx_;
  PA_REQUIRE(A_, "Operator must be set for GmresSolver::Mult!");
  const int n = A_->Height();
  const int m = (max_dim_ > 0) ? std::min(max_dim_, max_it_) : max_it_;
  if (r_.Size() != n) r_.SetSize(n);
  if ((int)V_.size() < m + 1) V_.resize(m + 1);
  auto ensure = [&](int j) {
    if (V_[j].Size() != n) V_[j].SetSize(n);
  };
  std::vector<cd> H((size_t)(m + 1) * m), s(m + 1), sn(m + 1);
  std::vector<double> cs(m + 1);
  auto Hij = [&](int i, int j) -> cd & { return H[(size_t)j * (m + 1) + i]; };
  bool have_guess = initial_guess;
  auto residual = [&]() {
    ensure(0);
    if (have_guess) {
      A_->Mult(x, r_);
      linalg::Scale(c, -1.0, r_);
      linalg::AXPY(c, cd(1.0, 0.0), b, r_);
    } else {
      linalg::Copy(c, b, r_);
      linalg::Fill(c, x, 0.0);
    }
    ApplyB(r_, V_[0]);
    return linalg::Norml2(c, V_[0]);
  };
  double beta = residual();
  if (initial_guess) {
    ensure(1);
    ApplyB(b, V_[1]);
    initial_res_ = linalg::Norml2(c, V_[1]);
  } else {
    initial_res_ = beta;
  }
  const double eps = std::max(rel_tol_ * initial_res_, abs_tol_);
  converged_ = beta < eps;
  double res = beta;
  int it = 0;
  while (it < max_it_ && !converged_ && beta > 0.0) {
    linalg::Scale(c, 1.0 / beta, V_[0]);
    std::fill(s.begin(), s.end(), cd(0.0));
    s[0] = beta;
    int j = 0;
    for (; j < m && it < max_it_; j++, it++) {
      ensure(j + 1);
      ComplexVector &w = V_[j + 1];
      A_->Mult(V_[j], r_);
      ApplyB(r_, w);
      for (int i = 0; i <= j; i++) {  // MGS, Dot(w, v_i) = v_i^H w
        Hij(i, j) = linalg::Dot(c, w, V_[i]);
        linalg::AXPY(c, -Hij(i, j), V_[i], w);
      }
      const double hn = linalg::Norml2(c, w);
      Hij(j + 1, j) = hn;
      if (hn != 0.0) linalg::Scale(c, 1.0 / hn, w);
      for (int k = 0; k < j; k++) {  // apply previous rotations [c s; -conj(s) c]
        const cd t = cs[k] * Hij(k, j) + sn[k] * Hij(k + 1, j);
        Hij(k + 1, j) = -std::conj(sn[k]) * Hij(k, j) + cs[k] * Hij(k + 1, j);
        Hij(k, j) = t;
      }
      {  // new rotation annihilating H(j+1, j)
        const cd f = Hij(j, j), g = Hij(j + 1, j);
        if (g == cd(0.0)) {
          cs[j] = 1.0, sn[j] = 0.0;
        } else if (f == cd(0.0)) {
          cs[j] = 0.0, sn[j] = std::conj(g) / std::abs(g);
        } else {
          const double nrm = std::sqrt(std::norm(f) + std::norm(g));
          cs[j] = std::abs(f) / nrm;
          sn[j] = (f / std::abs(f)) * std::conj(g) / nrm;
        }
        Hij(j, j) = cs[j] * f + sn[j] * g;
        Hij(j + 1, j) = 0.0;
        const cd t = cs[j] * s[j] + sn[j] * s[j + 1];
        s[j + 1] = -std::conj(sn[j]) * s[j] + cs[j] * s[j + 1];
        s[j] = t;
      }
      res = std::abs(s[j + 1]);
      if (print_ > 1) std::printf("  %3d (restart %d) KSP residual norm %.6e\n", it + 1, j + 1, res);
      converged_ = res < eps;
      if (converged_) {
        j++, it++;
        break;
      }
    }
    for (int i = j - 1; i >= 0; i--) {
      s[i] /= Hij(i, i);
      for (int k = i - 1; k >= 0; k--) s[k] -= Hij(k, i) * s[i];
    }
    for (int k = 0; k < j; k++) linalg::AXPY(c, s[k], V_[k], x);
    if (converged_) break;
    have_guess = true;
    beta = residual();
    res = beta;
    converged_ = beta < eps;
  }
  if (print_ > 0)
    std::printf("  GMRES (complex) %s in %d iterations (res %.3e, initial %.3e)\n",
                converged_ ? "converged" : "did NOT converge", it, res, initial_res_);
  final_res_ = res, final_it_ = it;
}

}  // namespace palace
