// Host check of the plane rotations of the GMRES family (palace_amd/csrc/krylov_impl.hpp: LAPACK's d/zlartg with safe scaling,
// what linalg/iterative.cpp:45-181 restates too): over 2 x 200 000 pairs spanning the whole exponent range, with zero real or
// imaginary parts and f = 0 mixed in, [c s; -conj(s) c] is unitary and annihilates g to rounding (evaluated in long double).
#include "krylov_impl.hpp"
#include <random>
using namespace palace::krylov;
int main() {
  std::mt19937_64 gen(1);
  std::uniform_real_distribution<double> U(-1, 1);
  std::uniform_int_distribution<int> E(-1000, 1000);
  double worst = 0;
  for (int it = 0; it < 200000; it++) {
    const double sf = std::ldexp(1.0, it % 3 ? E(gen) / (it % 7 + 1) : 0), sg = std::ldexp(1.0, it % 5 ? E(gen) / (it % 4 + 1) : 0);
    std::complex<double> f(U(gen) * sf, (it % 11 == 0 ? 0.0 : U(gen)) * sf), g((it % 13 == 0 ? 0.0 : U(gen)) * sg, U(gen) * sg);
    if (it % 17 == 0) f = 0;
    double c; std::complex<double> s;
    GeneratePlaneRotation(f, g, c, s);
    // checks in long double: unitarity and annihilation, relative to the larger input
    using LD = long double; using CL = std::complex<LD>;
    CL F(f.real(), f.imag()), G(g.real(), g.imag()), S(s.real(), s.imag());
    LD C = c;
    LD unit = std::abs(C * C + std::norm(S) - 1.0L);
    CL z = -std::conj(S) * F + C * G;
    LD scale = std::max(std::abs(F), std::abs(G));
    LD ann = scale > 0 ? std::abs(z) / scale : 0;
    worst = std::max<double>(worst, std::max<double>(unit, ann));
    // real version
    double cr, sr;
    GeneratePlaneRotation(f.real(), g.imag(), cr, sr);
    LD ur = std::abs((LD)cr * cr + (LD)sr * sr - 1.0L), zr = std::abs(-(LD)sr * f.real() + (LD)cr * g.imag());
    LD scr = std::max(std::abs((LD)f.real()), std::abs((LD)g.imag()));
    worst = std::max<double>(worst, std::max<double>(ur, scr > 0 ? zr / scr : 0));
  }
  std::printf("worst deviation %.3e\n", worst);
  return worst < 1e-15 ? 0 : 1;
}
