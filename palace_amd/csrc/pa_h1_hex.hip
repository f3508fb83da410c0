// H1 tensor-product hexahedra (diffusion, mass): placeholder until the kernels land.
#include "pa_internal.hpp"

namespace pa {
void launch_h1_hex_apply(const SubOp &, const double *, double *, hipStream_t) {
  throw Error("H1 hexahedron apply kernel not built yet");
}
void launch_h1_hex_diag(const SubOp &, double *, hipStream_t) {
  throw Error("H1 hexahedron diagonal kernel not built yet");
}
}  // namespace pa
