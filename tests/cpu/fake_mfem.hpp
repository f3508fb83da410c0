// Stand-in for the few MFEM types the Palace-side shim of INTEGRATION.md section 1 touches (mfem::Vector with a device
// copy, mfem::Operator, MFEM_ABORT / MFEM_VERIFY): test infrastructure only, so that the documented shim is COMPILED and RUN
// by tests/test_integration_shim.py.  Vector keeps a host array and a device mirror with MFEM's Read / Write / ReadWrite
// (on_device) validity protocol.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <utility>
#include <vector>

#define MFEM_ABORT(msg) (std::fprintf(stderr, "MFEM abort: %s\n", (msg)), std::exit(3))
#define MFEM_VERIFY(cond, msg) do { if (!(cond)) MFEM_ABORT(msg); } while (0)
using CeedIntScalar = union { int64_t i; double s; };  // 8-byte slots (fem/libceed/ceed.hpp)

namespace mfem {
class Vector {
  mutable std::vector<double> h_;
  mutable double *d_ = nullptr;
  mutable bool h_valid_ = true, d_valid_ = false;
  void ToDevice() const {
    if (!d_ && !h_.empty() && hipMalloc((void **)&d_, h_.size() * 8) != hipSuccess) MFEM_ABORT("hipMalloc");
    if (!d_valid_ && !h_.empty() && hipMemcpy(d_, h_.data(), h_.size() * 8, hipMemcpyHostToDevice) != hipSuccess) MFEM_ABORT("H2D");
    d_valid_ = true;
  }
  void ToHost() const {
    if (!h_valid_ && !h_.empty() && hipMemcpy(h_.data(), d_, h_.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) MFEM_ABORT("D2H");
    h_valid_ = true;
  }
  void Drop() { if (d_) (void)hipFree(d_); d_ = nullptr, d_valid_ = false, h_valid_ = true; }

public:
  Vector() = default;
  explicit Vector(int n) : h_(n, 0.0) {}
  Vector(const Vector &o) { o.ToHost(), h_ = o.h_; }
  Vector(Vector &&o) noexcept : h_(std::move(o.h_)), d_(o.d_), h_valid_(o.h_valid_), d_valid_(o.d_valid_) { o.d_ = nullptr, o.h_.clear(), o.h_valid_ = true, o.d_valid_ = false; }
  ~Vector() { Drop(); }
  Vector &operator=(const Vector &o) { o.ToHost(), Drop(), h_ = o.h_; return *this; }
  Vector &operator=(Vector &&o) noexcept { Drop(), h_ = std::move(o.h_), d_ = o.d_, h_valid_ = o.h_valid_, d_valid_ = o.d_valid_, o.d_ = nullptr, o.h_.clear(), o.h_valid_ = true, o.d_valid_ = false; return *this; }
  Vector &operator=(double s) { Drop(); for (double &v : h_) v = s; return *this; }
  int Size() const { return (int)h_.size(); }
  void SetSize(int n) { if (n != Size()) Drop(), h_.assign(n, 0.0); }
  const double *Read(bool on_dev = true) const { if (on_dev) { ToDevice(); return d_; } ToHost(); return h_.data(); }
  double *Write(bool on_dev = true) { if (on_dev) { ToDevice(), h_valid_ = false; return d_; } d_valid_ = false, h_valid_ = true; return h_.data(); }
  double *ReadWrite(bool on_dev = true) { if (on_dev) { ToHost(), ToDevice(), h_valid_ = false; return d_; } ToHost(), d_valid_ = false; return h_.data(); }
  const double *HostRead() const { return Read(false); }
  double *HostReadWrite() { return ReadWrite(false); }
  Vector &operator*=(const Vector &d) { double *a = HostReadWrite(); const double *b = d.HostRead(); for (int i = 0; i < Size(); i++) a[i] *= b[i]; return *this; }
  Vector &operator+=(const Vector &d) { double *a = HostReadWrite(); const double *b = d.HostRead(); for (int i = 0; i < Size(); i++) a[i] += b[i]; return *this; }
};
class Operator {
protected:
  int height, width;

public:
  Operator(int h, int w) : height(h), width(w) {}
  virtual ~Operator() = default;
  int Height() const { return height; }
  int Width() const { return width; }
  virtual void Mult(const Vector &x, Vector &y) const = 0;
  virtual void MultTranspose(const Vector &, Vector &) const { MFEM_ABORT("MultTranspose"); }
  virtual void AddMult(const Vector &, Vector &, const double = 1.0) const { MFEM_ABORT("AddMult"); }
  virtual void AddMultTranspose(const Vector &, Vector &, const double = 1.0) const { MFEM_ABORT("AddMultTranspose"); }
  virtual void AssembleDiagonal(Vector &) const { MFEM_ABORT("AssembleDiagonal"); }
};
}  // namespace mfem
namespace palace {
using Vector = mfem::Vector;
using Operator = mfem::Operator;
}  // namespace palace
