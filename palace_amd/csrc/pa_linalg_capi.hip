// C ABI over the C++ host layer (include/palace_amd_linalg.h).
#include <memory>

#include "../../include/palace_amd_linalg.h"
#include "comm.hpp"
#include "complex.hpp"
#include "amg_dist.hpp"
#include "amg_solver.hpp"
#include "ksp.hpp"
#include "linalg.hpp"

using namespace palace;

namespace palace {
Operator *make_dense_interp_operator(const Context &ctx, const pa_restriction_desc &rd, const pa_restriction_desc &rr,
                                     const double *M, const Halo *halo_d, int nt_d, int nt_r, int nmat = 1,
                                     const uint8_t *mat_id = nullptr);
Operator *make_interp_operator(const Context &ctx, const pa_restriction_desc &rc, const pa_basis_desc &bc,
                               const pa_restriction_desc &rf, const pa_basis_desc &bf, const double *Ic,
                               const double *Io, const Halo *halo_c, int nt_c, int nt_f, int kind);
}

struct pa_context {
  Context ctx;
  std::unique_ptr<Comm> comm;
  hipStream_t own_stream = nullptr;
  ~pa_context() {
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};
struct pa_halo {
  std::unique_ptr<Halo> halo;
};
struct pa_par_op {
  pa_context *ctx;
  std::unique_ptr<ceed::Operator> local;
  std::vector<std::unique_ptr<ceed::Operator>> terms;  // BuildParSumOperator: the local operators of the sum
  std::unique_ptr<SumOperator> sum;
  std::unique_ptr<CsrOperator> csr;  // assembled local operator (coarsest level)
  std::unique_ptr<ParOperator> op;
};
struct pa_interp {
  pa_context *ctx;
  std::unique_ptr<Operator> op;
};
struct pa_solver {
  pa_context *ctx;
  std::unique_ptr<Solver> solver;
  std::vector<pa_solver *> owned;  // sub-solvers whose lifetime is tied to this one
  ~pa_solver() {
    for (auto *s : owned) delete s;
  }
};

struct pa_complex_par_op {
  pa_context *ctx;
  std::unique_ptr<ceed::Operator> local_r, local_i;
  std::unique_ptr<ComplexParOperator> op;
};

struct pa_csolver {
  pa_context *ctx;
  std::unique_ptr<ComplexWrapperOperator> A;  // owned wrapper (created from two real ParOperators), or
  const ComplexOperator *Aext = nullptr;      // a ComplexParOperator owned elsewhere
  std::unique_ptr<ComplexIterativeSolver> solver;
  int Height() const { return A ? A->Height() : Aext->Height(); }
};

using pa::guarded;

extern "C" {

int pa_context_create(void *stream, pa_context **ctx) {
  return guarded([&] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
      throw pa::Error("no HIP device visible: libpalace_amd has no CPU fallback");
    auto c = std::make_unique<pa_context>();
    if (stream) {
      c->ctx.stream = (hipStream_t)stream;
    } else {
      // a stream of the context's own with the default (blocking) flag: ordered against the legacy null stream like the
      // null stream itself, so callers that fill vectors there (PyTorch's default stream) see the same ordering, and --
      // unlike the null stream -- it can be recorded into HIP graphs
      PA_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamDefault));
      c->ctx.stream = c->own_stream;
    }
    *ctx = c.release();
  });
}
void pa_context_destroy(pa_context *ctx) { delete ctx; }
int pa_context_stream(const pa_context *ctx, void **stream) {
  return guarded([&] {
    PA_REQUIRE(ctx && stream, "null argument");
    *stream = (void *)ctx->ctx.stream;
  });
}
int pa_context_synchronize(pa_context *ctx) {
  return guarded([&] {
    PA_HIP(hipStreamSynchronize(ctx->ctx.stream));
    if (ctx->comm) ctx->comm->PeerCheckNow();
  });
}

int pa_comm_unique_id(char *out128) {
  return guarded([&] { Comm::GetUniqueId(out128); });
}
int pa_context_init_comm(pa_context *ctx, int rank, int size, const char *id) {
  return guarded([&] {
    PA_REQUIRE(ctx && id && size >= 1 && rank >= 0 && rank < size, "bad communicator arguments");
    ctx->comm = std::make_unique<Comm>(rank, size, id);
    ctx->ctx.comm = ctx->comm.get();
  });
}
struct pa_local_group {
  LocalGroup group;
  explicit pa_local_group(int size) : group(size) {}
};
int pa_local_group_create(int size, pa_local_group **out) {
  return guarded([&] {
    PA_REQUIRE(size >= 1 && out, "bad group size");
    *out = new pa_local_group(size);
  });
}
void pa_local_group_destroy(pa_local_group *g) { delete g; }
void pa_local_group_abort(pa_local_group *g) {
  if (g) g->group.Abort();
}
int pa_context_init_comm_local(pa_context *ctx, int rank, pa_local_group *g) {
  return guarded([&] {
    PA_REQUIRE(ctx && g && rank >= 0 && rank < g->group.Size(), "bad communicator arguments");
    ctx->comm = std::make_unique<Comm>(rank, g->group);
    ctx->ctx.comm = ctx->comm.get();
  });
}
int pa_context_init_comm_peer(pa_context *ctx, int rank, int size) {
  return guarded([&] {
    PA_REQUIRE(ctx && size >= 1 && rank >= 0 && rank < size, "bad communicator arguments");
    ctx->comm = std::make_unique<Comm>(rank, size);
    ctx->ctx.comm = ctx->comm.get();
  });
}
int pa_comm_peer_handle(pa_context *ctx, char *out64) {
  return guarded([&] {
    PA_REQUIRE(ctx && ctx->comm && out64, "no communicator");
    ctx->comm->PeerHandle(out64);
  });
}
int pa_comm_peer_connect(pa_context *ctx, const char *handles) {
  return guarded([&] {
    PA_REQUIRE(ctx && ctx->comm && handles, "no communicator");
    ctx->comm->PeerConnect(handles);
  });
}
int pa_comm_peer_disconnect(pa_context *ctx) {
  return guarded([&] {
    PA_REQUIRE(ctx && ctx->comm, "no communicator");
    ctx->comm->PeerDisconnect();
  });
}
int pa_comm_peer_ready(const pa_context *ctx) { return ctx && ctx->comm && ctx->comm->PeerReady() ? 1 : 0; }
int pa_comm_peer_check(pa_context *ctx) {
  return guarded([&] {
    if (ctx && ctx->comm) ctx->comm->PeerCheck(ctx->ctx.stream);
  });
}
int pa_comm_peer_set_fenced(int on) {
  return guarded([&] { Comm::SetFenced(on != 0); });
}
int pa_comm_peer_fenced(void) { return Comm::Fenced() ? 1 : 0; }
int pa_comm_peer_set_timeout(double seconds) {
  return guarded([&] { Comm::SetTimeout(seconds); });
}
int pa_comm_peer_stress(pa_context *ctx, pa_halo *ring, int n, int rounds, int direct, int graph, long long *failures) {
  return guarded([&] {
    PA_REQUIRE(ctx && ctx->comm && ring && failures && n > 0 && rounds > 0, "bad arguments");
    *failures = ctx->comm->StressRing(*ring->halo, n, rounds, direct != 0, graph != 0, ctx->ctx.stream);
  });
}
int pa_halo_uses_peer(const pa_halo *halo) { return halo && halo->halo->UsesPeerTransport() ? 1 : 0; }
int pa_context_rank(const pa_context *ctx) { return ctx && ctx->comm ? ctx->comm->Rank() : 0; }
int pa_context_size(const pa_context *ctx) { return ctx && ctx->comm ? ctx->comm->Size() : 1; }
int pa_allreduce_sum(pa_context *ctx, double *buf, int n) {
  return guarded([&] {
    if (ctx->comm) ctx->comm->AllReduceSum(buf, n, ctx->ctx.stream);
  });
}

int pa_halo_create(pa_context *ctx, int nnbr, const int *nbr, const int *send_off, const int32_t *send_idx,
                   const int *recv_off, const int32_t *recv_idx, pa_halo **halo) {
  return guarded([&] {
    PA_REQUIRE(ctx && ctx->comm, "halo plan needs a communicator (pa_context_init_comm)");
    auto *h = new pa_halo;
    h->halo = std::make_unique<Halo>(*ctx->comm, nnbr, nbr, send_off, send_idx, recv_off, recv_idx);
    *halo = h;
  });
}
void pa_halo_destroy(pa_halo *halo) { delete halo; }
int pa_halo_prolongate(pa_context *ctx, pa_halo *halo, double *lx) {
  return guarded([&] { halo->halo->Prolongate(lx, ctx->ctx.stream); });
}
int pa_halo_restrict_add(pa_context *ctx, pa_halo *halo, double *ly) {
  return guarded([&] { halo->halo->RestrictAdd(ly, ctx->ctx.stream); });
}

int pa_par_op_create(pa_context *ctx, pa_op *local, int n_true, const int32_t *ess, int n_ess, int policy,
                     pa_halo *halo, pa_par_op **A) {
  return guarded([&] {
    PA_REQUIRE(ctx && local && A, "null argument");
    auto *p = new pa_par_op;
    p->ctx = ctx;
    p->local = std::make_unique<ceed::Operator>(ctx->ctx, local, false);
    p->op = std::make_unique<ParOperator>(ctx->ctx, *p->local, n_true, ess, n_ess,
                                          policy == PA_DIAG_ONE ? ParOperator::DiagonalPolicy::DIAG_ONE
                                                                : ParOperator::DiagonalPolicy::DIAG_ZERO,
                                          halo ? halo->halo.get() : nullptr);
    *A = p;
  });
}
int pa_par_op_create_assembled(pa_context *ctx, const pa_csr *csr, int n_true, const int32_t *ess, int n_ess,
                               int policy, pa_halo *halo, pa_par_op **A) {
  return guarded([&] {
    PA_REQUIRE(ctx && csr && A, "null argument");
    auto *p = new pa_par_op;
    p->ctx = ctx;
    p->csr = std::make_unique<CsrOperator>(ctx->ctx, csr);
    p->op = std::make_unique<ParOperator>(ctx->ctx, *p->csr, n_true, ess, n_ess,
                                          policy == PA_DIAG_ONE ? ParOperator::DiagonalPolicy::DIAG_ONE
                                                                : ParOperator::DiagonalPolicy::DIAG_ZERO,
                                          halo ? halo->halo.get() : nullptr);
    *A = p;
  });
}
int pa_par_sum_op_create(pa_context *ctx, int nterms, pa_op *const *locals, const double *coeffs, int n_true,
                         const int32_t *ess, int n_ess, int policy, pa_halo *halo, pa_par_op **A) {
  return guarded([&] {
    PA_REQUIRE(ctx && locals && coeffs && A && nterms >= 1, "bad argument");
    auto *p = new pa_par_op;
    p->ctx = ctx;
    const int n = pa_op_height(locals[0]);
    p->sum = std::make_unique<SumOperator>(ctx->ctx, n, n);
    for (int k = 0; k < nterms; k++) {
      PA_REQUIRE(locals[k], "null local operator");
      p->terms.push_back(std::make_unique<ceed::Operator>(ctx->ctx, locals[k], false));
      p->sum->AddOperator(*p->terms.back(), coeffs[k]);
    }
    p->op = std::make_unique<ParOperator>(ctx->ctx, *p->sum, n_true, ess, n_ess,
                                          policy == PA_DIAG_ONE ? ParOperator::DiagonalPolicy::DIAG_ONE
                                                                : ParOperator::DiagonalPolicy::DIAG_ZERO,
                                          halo ? halo->halo.get() : nullptr);
    *A = p;
  });
}
void pa_par_op_destroy(pa_par_op *A) { delete A; }
int pa_par_op_direct_form(const pa_par_op *A) { return (A && A->op) ? A->op->DirectForm() : -1; }
int pa_par_op_set_direct(pa_par_op *A, int on) {
  return guarded([&] {
    PA_REQUIRE(A && A->op, "null argument");
    A->op->SetDirect(on != 0);
  });
}
int pa_par_op_mult(pa_par_op *A, const double *x, double *y) {
  return guarded([&] {
    Vector vx(const_cast<double *>(x), A->op->Width()), vy(y, A->op->Height());
    A->op->Mult(vx, vy);
  });
}
int pa_par_op_add_mult(pa_par_op *A, const double *x, double *y, double a) {
  return guarded([&] {
    PA_REQUIRE(A && x && y, "null argument");
    Vector vx(const_cast<double *>(x), A->op->Width()), vy(y, A->op->Height());
    A->op->AddMult(vx, vy, a);
  });
}
int pa_par_op_eliminate_rhs(pa_par_op *A, const double *x, double *b) {
  return guarded([&] {
    PA_REQUIRE(A && x && b, "null argument");
    Vector vx(const_cast<double *>(x), A->op->Width()), vb(b, A->op->Height());
    A->op->EliminateRHS(vx, vb);
  });
}
int pa_par_op_assemble_diagonal(pa_par_op *A, double *diag) {
  return guarded([&] {
    Vector d(diag, A->op->Height());
    A->op->AssembleDiagonal(d);
  });
}

int pa_vec_dot(pa_context *ctx, const double *x, const double *y, int n, double *result) {
  return guarded([&] {
    Vector vx(const_cast<double *>(x), n), vy(const_cast<double *>(y), n);
    *result = linalg::Dot(ctx->ctx, vx, vy);
  });
}
int pa_vec_sum(pa_context *ctx, const double *x, int n, double *result) {
  return guarded([&] {
    PA_REQUIRE(ctx && result && n >= 0, "bad argument");
    Vector vx(const_cast<double *>(x), n);
    *result = linalg::Sum(ctx->ctx, vx);
  });
}
int pa_vec_sqrt(pa_context *ctx, double *x, int n, double s) {
  return guarded([&] {
    PA_REQUIRE(ctx && n >= 0, "bad argument");
    Vector vx(x, n);
    linalg::Sqrt(ctx->ctx, vx, s);
  });
}
int pa_vec_axpby(pa_context *ctx, double a, const double *x, double b, double *y, int n) {
  return guarded([&] {
    Vector vx(const_cast<double *>(x), n), vy(y, n);
    linalg::AXPBY(ctx->ctx, a, vx, b, vy);
  });
}
/* ComplexVector members (linalg/vector.hpp:95-146) on split real / imaginary arrays; op: 0 x *= a, 1 x = |x|, 2 x = 1 ./ x,
 * 3 x = conj(x), 4 y = a x + b y, 5 z = a x + b y + c z, 6 out = x^T y (no conjugate) */
int pa_cvec_op(pa_context *ctx, int op, int n, const double *coef, double *xr, double *xi, double *yr, double *yi, double *zr,
               double *zi, double *out) {
  return guarded([&] {
    PA_REQUIRE(ctx && n >= 0 && xr && xi, "bad argument");
    using cd = std::complex<double>;
    ComplexVector x(xr, xi, n), y(yr, yi, yr ? n : 0), z(zr, zi, zr ? n : 0);
    auto co = [&](int k) { return coef ? cd(coef[2 * k], coef[2 * k + 1]) : cd(1.0); };
    switch (op) {
      case 0: linalg::Scale(ctx->ctx, co(0), x); break;
      case 1: linalg::Abs(ctx->ctx, x); break;
      case 2: linalg::Reciprocal(ctx->ctx, x); break;
      case 3: linalg::Conj(ctx->ctx, x); break;
      case 4: PA_REQUIRE(yr && yi, "y missing"); linalg::AXPBY(ctx->ctx, co(0), x, co(1), y); break;
      case 5: PA_REQUIRE(yr && yi && zr && zi, "y / z missing"); linalg::AXPBYPCZ(ctx->ctx, co(0), x, co(1), y, co(2), z); break;
      case 6: {
        PA_REQUIRE(yr && yi && out, "y / out missing");
        const cd d = linalg::TransposeDot(ctx->ctx, x, y);
        out[0] = d.real(), out[1] = d.imag();
      } break;
      default: throw pa::Error("unknown complex vector operation");
    }
  });
}
/* ComplexVector::SetBlocks (vector.cpp:172-201): x = [s_0 y_0; s_1 y_1; ...]; s NULL = ones */
int pa_cvec_set_blocks(pa_context *ctx, double *xr, double *xi, int n, int nblocks, const double *const *yr,
                       const double *const *yi, const int *sizes, const double *s) {
  return guarded([&] {
    PA_REQUIRE(ctx && xr && xi && nblocks >= 0, "bad argument");
    ComplexVector x(xr, xi, n);
    std::vector<ComplexVector> blocks;
    blocks.reserve((size_t)nblocks);
    std::vector<const ComplexVector *> ptr;
    std::vector<std::complex<double>> sc;
    for (int b = 0; b < nblocks; b++) {
      blocks.emplace_back(const_cast<double *>(yr[b]), const_cast<double *>(yi[b]), sizes[b]);
      if (s) sc.emplace_back(s[2 * b], s[2 * b + 1]);
    }
    for (auto &b : blocks) ptr.push_back(&b);
    linalg::SetBlocks(ctx->ctx, x, ptr, sc);
  });
}
/* DiagonalOperator / ComplexDiagonalOperator (linalg/operator.hpp:354-423): y (+)= a op(diag(d)) x; mode 0 N, 1 T, 2 H
 * (imaginary arrays NULL: the real operator) */
int pa_diag_op_apply(pa_context *ctx, int n, const double *dr, const double *di, const double *xr, const double *xi, double *yr,
                     double *yi, double ar, double ai, int mode, int add) {
  return guarded([&] {
    PA_REQUIRE(ctx && dr && xr && yr, "bad argument");
    if (!di) {
      Vector d(const_cast<double *>(dr), n), x(const_cast<double *>(xr), n), y(yr, n);
      DiagonalOperator D(ctx->ctx, d);
      if (add) mode ? D.AddMultTranspose(x, y, ar) : D.AddMult(x, y, ar); else mode ? D.MultTranspose(x, y) : D.Mult(x, y);
      return;
    }
    PA_REQUIRE(xi && yi, "imaginary parts missing");
    ComplexVector d(const_cast<double *>(dr), const_cast<double *>(di), n), x(const_cast<double *>(xr), const_cast<double *>(xi), n),
        y(yr, yi, n);
    ComplexDiagonalOperator D(ctx->ctx, d);
    const std::complex<double> a(ar, ai);
    if (add)
      mode == 0 ? D.AddMult(x, y, a) : mode == 1 ? D.AddMultTranspose(x, y, a) : D.AddMultHermitianTranspose(x, y, a);
    else
      mode == 0 ? D.Mult(x, y) : mode == 1 ? D.MultTranspose(x, y) : D.MultHermitianTranspose(x, y);
  });
}
/* ProductOperator (linalg/operator.hpp:270-352): y (+)= a op(A B) x over two ParOperators */
int pa_product_op_apply(pa_par_op *A, pa_par_op *B, const double *x, double *y, int transpose, double a, int add) {
  return guarded([&] {
    PA_REQUIRE(A && B && x && y, "null argument");
    ProductOperator AB(*A->op, *B->op);
    Vector vx(const_cast<double *>(x), transpose ? AB.Height() : AB.Width()), vy(y, transpose ? AB.Width() : AB.Height());
    if (add) transpose ? AB.AddMultTranspose(vx, vy, a) : AB.AddMult(vx, vy, a); else transpose ? AB.MultTranspose(vx, vy) : AB.Mult(vx, vy);
  });
}
/* ComplexProductOperator over two ComplexParOperators: mode 0 N, 1 T, 2 H */
int pa_complex_product_op_apply(pa_complex_par_op *A, pa_complex_par_op *B, const double *xr, const double *xi, double *yr,
                                double *yi, int mode, double ar, double ai, int add) {
  return guarded([&] {
    PA_REQUIRE(A && B && xr && xi && yr && yi, "null argument");
    ComplexProductOperator AB(*A->op, *B->op);
    const int n = AB.Height();
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), n), y(yr, yi, n);
    const std::complex<double> a(ar, ai);
    if (add)
      mode == 0 ? AB.AddMult(x, y, a) : mode == 1 ? AB.AddMultTranspose(x, y, a) : AB.AddMultHermitianTranspose(x, y, a);
    else
      mode == 0 ? AB.Mult(x, y) : mode == 1 ? AB.MultTranspose(x, y) : AB.MultHermitianTranspose(x, y);
  });
}
// Measured FP64 matrix-core peak (bench.py's denominator beside the data-sheet figure): every wave
// keeps eight independent 16x16 accumulators and issues `iters` rounds of v_mfma_f64_16x16x4_f64 on them.
typedef double pa_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(const int iters, double *__restrict__ out) {
  pa_d4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = pa_d4{0.0, 0.0, 0.0, 0.0};
  double a = 1e-3 * (threadIdx.x & 63), b = 1.0 + 1e-6 * (threadIdx.x & 15);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == -1.0) out[0] = s;  // keeps the chain alive, never true
}
int pa_bench_mfma_f64(pa_context *ctx, int iters, int n_blocks, double *scratch, double *flops_per_launch) {
  return guarded([&] {
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(n_blocks), dim3(256), 0, ctx->ctx.stream, iters, scratch);
    PA_HIP(hipGetLastError());
    // one MFMA = 16 x 16 x 4 multiply-adds; 4 waves per block, 8 per round
    *flops_per_launch = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 * n_blocks;
  });
}
int pa_vec_set_random(pa_context *ctx, double *x, int n, uint64_t seed) {
  return guarded([&] {
    Vector vx(x, n);
    linalg::SetRandom(ctx->ctx, vx, seed);
  });
}

int pa_chebyshev_create(pa_context *ctx, pa_par_op *A, int smooth_it, int order, double sf_max, int fourth_kind,
                        pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(order > 0, "Polynomial order for Chebyshev smoothing must be positive!");
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto cheb = std::make_unique<ChebyshevSmoother>(ctx->ctx, smooth_it, order, sf_max, fourth_kind != 0);
    cheb->SetOperator(*A->op);
    s->solver = std::move(cheb);
    *S = s;
  });
}
/* ChebyshevSmoother1stKind (linalg/chebyshev.cpp:222-293); sf_min <= 0 takes the optimised estimate (:244-247) */
int pa_chebyshev_create_1st_kind(pa_context *ctx, pa_par_op *A, int smooth_it, int order, double sf_max, double sf_min,
                                 pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(order > 0, "Polynomial order for Chebyshev smoothing must be positive!");
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto cheb = std::make_unique<ChebyshevSmoother>(ctx->ctx, smooth_it, order, sf_max, false, sf_min);
    cheb->SetOperator(*A->op);
    s->solver = std::move(cheb);
    *S = s;
  });
}
/* DistRelaxationSmoother (linalg/distrelaxation.cpp:14-151) on its own: A the Nedelec ParOperator, A_aux the auxiliary
 * H1 ParOperator, G the discrete gradient */
int pa_dist_relaxation_create(pa_context *ctx, pa_par_op *A, pa_par_op *A_aux, pa_interp *G, int smooth_it,
                              int cheby_smooth_it, int cheby_order, double sf_max, double sf_min, int fourth_kind,
                              pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && A_aux && G && S, "null argument");
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto d = std::make_unique<DistRelaxationSmoother>(ctx->ctx, *G->op, smooth_it, cheby_smooth_it, cheby_order, sf_max,
                                                      sf_min, fourth_kind != 0);
    d->SetOperators(*A->op, *A_aux->op);
    s->solver = std::move(d);
    *S = s;
  });
}
int pa_dist_relaxation_lambda_max(const pa_solver *S, double *lambda_max, double *lambda_max_aux) {
  return guarded([&] {
    auto *d = dynamic_cast<const DistRelaxationSmoother *>(S ? S->solver.get() : nullptr);
    PA_REQUIRE(d && lambda_max && lambda_max_aux, "not a distributive relaxation smoother");
    *lambda_max = d->Primary().LambdaMax(), *lambda_max_aux = d->Auxiliary().LambdaMax();
  });
}
/* Solver::Mult2 / MultTranspose2 (the entry points of the V-cycle, gmg.cpp:184,204): y <- y + B (x - A y) */
int pa_solver_mult2(pa_solver *S, const double *x, double *y, int transpose, int initial_guess) {
  return guarded([&] {
    PA_REQUIRE(S && x && y, "null argument");
    const int n = S->solver->Height();
    Vector vx(const_cast<double *>(x), n), vy(y, n), r(n);
    S->solver->SetInitialGuess(initial_guess != 0);
    if (transpose)
      S->solver->MultTranspose2(vx, vy, r);
    else
      S->solver->Mult2(vx, vy, r);
  });
}
/* CgSolver: iterations the host may run ahead of the last residual it has read (default 1; < 0 never waits);
 * host_scalars != 0 selects the reference's synchronous loop */
int pa_cg_set_lookahead(pa_solver *S, int lookahead, int host_scalars) {
  return guarded([&] {
    auto *k = dynamic_cast<CgSolver *>(S ? S->solver.get() : nullptr);
    PA_REQUIRE(k, "not a CG solver");
    k->SetLookahead(lookahead);
    k->SetHostScalars(host_scalars != 0);
  });
}
/* eigenvalue estimate of the Chebyshev smoother of multigrid level l >= 1 (plain Chebyshev levels only) */
int pa_gmg_smoother_lambda_max(const pa_solver *S, int level, double *lambda_max) {
  return guarded([&] {
    auto *g = dynamic_cast<const GeometricMultigridSolver *>(S ? S->solver.get() : nullptr);
    PA_REQUIRE(g && lambda_max, "not a multigrid solver");
    auto *c = dynamic_cast<const ChebyshevSmoother *>(&g->Smoother(level));
    PA_REQUIRE(c, "level smoother is not a Chebyshev smoother");
    *lambda_max = c->LambdaMax();
  });
}
/* 1 when the Chebyshev smoother (level >= 1 of a multigrid solver, or the solver itself with level < 0) runs its steps inside
 * the operator's E^T (Operator::MultChebyStep, pa_op_mult_cheb_step) */
int pa_chebyshev_fused_step(const pa_solver *S, int level, int *fused) {
  return guarded([&] {
    PA_REQUIRE(S && S->solver && fused, "null argument");
    const ChebyshevSmoother *c = nullptr;
    if (level < 0) {
      c = dynamic_cast<const ChebyshevSmoother *>(S->solver.get());
    } else {
      auto *g = dynamic_cast<const GeometricMultigridSolver *>(S->solver.get());
      PA_REQUIRE(g, "not a multigrid solver");
      c = dynamic_cast<const ChebyshevSmoother *>(&g->Smoother(level));
    }
    PA_REQUIRE(c, "not a Chebyshev smoother");
    *fused = c->FusedStep() ? 1 : 0;
  });
}
int pa_chebyshev_lambda_max(const pa_solver *S, double *lambda_max) {
  return guarded([&] {
    auto *c = dynamic_cast<const ChebyshevSmoother *>(S->solver.get());
    PA_REQUIRE(c, "not a Chebyshev smoother");
    *lambda_max = c->LambdaMax();
  });
}
int pa_replicated_solver_create(pa_context *ctx, pa_halo *gather, pa_solver *inner, const int32_t *mine, const double *sign,
                                int n_true, int n_global, pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && gather && inner && S && (mine || n_true == 0), "null argument");
    auto *s = new pa_solver;
    s->ctx = ctx;
    s->solver = std::make_unique<ReplicatedSolver>(ctx->ctx, *gather->halo, *inner->solver, mine, n_true, n_global, sign);
    *S = s;
  });
}
int pa_jacobi_create(pa_context *ctx, pa_par_op *A, pa_solver **S) {
  return guarded([&] {
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto j = std::make_unique<JacobiSmoother>(ctx->ctx);
    j->SetOperator(*A->op);
    s->solver = std::move(j);
    *S = s;
  });
}
static AmgOptions amg_options(const pa_amg_options *o) {
  AmgOptions a;
  if (o) {
    if (o->max_levels > 0) a.max_levels = o->max_levels;
    if (o->coarse_size > 0) a.coarse_size = o->coarse_size;
    if (o->smooth_order > 0) a.smooth_order = o->smooth_order;
    if (o->theta > 0.0) a.theta = o->theta;
  }
  return a;
}
int pa_amg_create(pa_context *ctx, const pa_csr *A, const int32_t *ess, int n_ess, const pa_amg_options *opt, pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && S && (ess || n_ess == 0), "null argument");
    auto *s = new pa_solver;
    s->ctx = ctx;
    s->solver = std::make_unique<AmgSolver>(ctx->ctx, DownloadCsr(*A, ess, n_ess), amg_options(opt));
    *S = s;
  });
}
int pa_ams_create(pa_context *ctx, const pa_csr *A, const int32_t *ess, int n_ess, int n_vert, const int32_t *G_rowptr,
                  const int32_t *G_col, const double *G_val, const double *coords, int dim, const pa_ams_options *opt,
                  pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && S && G_rowptr && G_col && G_val && coords && (ess || n_ess == 0), "null argument");
    amg::HostCsr G;
    G.nrows = A->nrows, G.ncols = n_vert;
    G.rowptr.assign(G_rowptr, G_rowptr + A->nrows + 1);
    G.col.assign(G_col, G_col + G.rowptr.back());
    G.val.assign(G_val, G_val + G.rowptr.back());
    for (int c : G.col) PA_REQUIRE(c >= 0 && c < n_vert, "discrete gradient column out of range");
    std::vector<char> flag((size_t)A->nrows, 0);
    for (int i = 0; i < n_ess; i++) {
      PA_REQUIRE(ess[i] >= 0 && ess[i] < A->nrows, "essential dof out of range");
      flag[ess[i]] = 1;
    }
    AmsOptions o;
    if (opt) {
      if (opt->cycle_it > 0) o.cycle_it = opt->cycle_it;
      if (opt->smooth_order > 0) o.smooth_order = opt->smooth_order;
      o.singular = opt->singular != 0;
      o.amg = amg_options(&opt->amg);
    }
    auto *s = new pa_solver;
    s->ctx = ctx;
    s->solver = std::make_unique<AmsSolver>(ctx->ctx, DownloadCsr(*A, ess, n_ess), G, coords, dim, flag, o);
    *S = s;
  });
}
int pa_replicated_coarse_create(pa_context *ctx, pa_par_op *level0, pa_interp *G, int nv_true, const double *xyz_true, int dim,
                                int cycle_it, int singular, pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && level0 && level0->op && S && (!G || (G->op && xyz_true)), "null argument");
    auto s = std::make_unique<pa_solver>();  // (released only once the constructor below has not thrown)
    s->ctx = ctx;
    s->solver = std::make_unique<ReplicatedCoarseSolver>(ctx->ctx, *level0->op, G ? G->op.get() : nullptr, nv_true, xyz_true, dim,
                                                         cycle_it, singular != 0);
    *S = s.release();
  });
}
int pa_replicated_coarse_info(const pa_solver *S, int *distributed, int *levels) {
  return guarded([&] {
    PA_REQUIRE(S && S->solver && distributed, "null argument");
    const auto *r = dynamic_cast<const ReplicatedCoarseSolver *>(S->solver.get());
    PA_REQUIRE(r, "not a multi-rank coarse solver (pa_replicated_coarse_create)");
    *distributed = r->Distributed() ? 1 : 0;
    if (levels) {
      *levels = 0;
      if (const auto *a = dynamic_cast<const DistAmgSolver *>(r->DistributedSolver())) *levels = a->NumLevels();
      if (const auto *m = dynamic_cast<const DistAmsSolver *>(r->DistributedSolver())) *levels = m->NodalSpaceSolver()->NumLevels();
    }
  });
}
static const AmgSolver &amg_of(const pa_solver *S, int which) {
  PA_REQUIRE(S && S->solver, "null solver");
  if (which == 0) {
    auto *a = dynamic_cast<const AmgSolver *>(S->solver.get());
    PA_REQUIRE(a, "not an AMG solver");
    return *a;
  }
  auto *m = dynamic_cast<const AmsSolver *>(S->solver.get());
  PA_REQUIRE(m && (which == 1 || which == 2), "not an AMS solver / unknown component");
  const AmgSolver *a = which == 1 ? m->GradientSpaceSolver() : m->NodalSpaceSolver();
  PA_REQUIRE(a, "this AMS solver has no such component (singular operator: no gradient-space solver)");
  return *a;
}
int pa_amg_num_levels(const pa_solver *S, int which, int *nlevels) {
  return guarded([&] {
    PA_REQUIRE(nlevels, "null argument");
    *nlevels = amg_of(S, which).NumLevels();
  });
}
int pa_amg_get_matrix(const pa_solver *S, int which, int level, int kind, int32_t *nrows, int32_t *ncols, int64_t *nnz,
                      int32_t *rowptr, int32_t *col, double *val) {
  return guarded([&] {
    const AmgSolver &a = amg_of(S, which);
    const amg::Hierarchy &h = a.HostHierarchy();
    if (kind == 2) {
      const int n = h.A.back().nrows;
      PA_REQUIRE(!a.HostCoarseInverse().empty() || n == 0, "no direct solve on the last level of this hierarchy");
      if (nrows) *nrows = n;
      if (ncols) *ncols = n;
      if (nnz) *nnz = (int64_t)n * n;
      if (val) std::copy(a.HostCoarseInverse().begin(), a.HostCoarseInverse().end(), val);
      return;
    }
    PA_REQUIRE(kind == 0 || kind == 1, "unknown matrix kind");
    PA_REQUIRE(level >= 0 && level < (int)(kind == 0 ? h.A.size() : h.P.size()), "level out of range");
    const amg::HostCsr &m = kind == 0 ? h.A[level] : h.P[level];
    if (nrows) *nrows = m.nrows;
    if (ncols) *ncols = m.ncols;
    if (nnz) *nnz = m.nnz();
    if (rowptr) std::copy(m.rowptr.begin(), m.rowptr.end(), rowptr);
    if (col) std::copy(m.col.begin(), m.col.end(), col);
    if (val) std::copy(m.val.begin(), m.val.end(), val);
  });
}
static void configure(IterativeSolver &k, pa_par_op *A, pa_solver *pc, double rel, double abs, int max_it) {
  k.SetOperator(*A->op);
  if (pc) k.SetPreconditioner(*pc->solver);
  k.SetTol(rel), k.SetAbsTol(abs), k.SetMaxIter(max_it);
}
int pa_cg_create(pa_context *ctx, pa_par_op *A, pa_solver *pc, double rel, double abs, int max_it, int print,
                 pa_solver **S) {
  return guarded([&] {
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto k = std::make_unique<CgSolver>(ctx->ctx, print);
    configure(*k, A, pc, rel, abs, max_it);
    s->solver = std::move(k);
    *S = s;
  });
}
int pa_gmres_create(pa_context *ctx, pa_par_op *A, pa_solver *pc, double rel, double abs, int max_it, int restart,
                    int flexible, int print, pa_solver **S) {
  return guarded([&] {
    auto *s = new pa_solver;
    s->ctx = ctx;
    auto k = std::make_unique<GmresSolver>(ctx->ctx, print, flexible != 0);
    configure(*k, A, pc, rel, abs, max_it);
    k->SetRestartDim(restart);
    s->solver = std::move(k);
    *S = s;
  });
}
static void gmg_create(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P, pa_par_op *const *A_aux,
                       pa_interp *const *G, pa_solver *coarse, int cycle_it, int smooth_it, int cheby_order,
                       double sf_max, double sf_min, int fourth, pa_solver **S);

int pa_gmg_create(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P, pa_solver *coarse,
                  int cycle_it, int smooth_it, int cheby_order, double sf_max, double sf_min, int fourth,
                  pa_solver **S) {
  return guarded([&] {
    gmg_create(ctx, nlevels, A, P, nullptr, nullptr, coarse, cycle_it, smooth_it, cheby_order, sf_max, sf_min, fourth, S);
  });
}
int pa_gmg_create_aux(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P, pa_par_op *const *A_aux,
                      pa_interp *const *G, pa_solver *coarse, int cycle_it, int smooth_it, int cheby_order,
                      double sf_max, double sf_min, int fourth, pa_solver **S) {
  return guarded([&] {
    PA_REQUIRE(A_aux && G, "auxiliary operators and discrete gradients are required");
    gmg_create(ctx, nlevels, A, P, A_aux, G, coarse, cycle_it, smooth_it, cheby_order, sf_max, sf_min, fourth, S);
  });
}

static void gmg_create(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P, pa_par_op *const *A_aux,
                       pa_interp *const *G, pa_solver *coarse, int cycle_it, int smooth_it, int cheby_order,
                       double sf_max, double sf_min, int fourth, pa_solver **S) {
  {
    PA_REQUIRE(nlevels >= 1 && A && coarse, "Empty finite element space hierarchy during multigrid solver setup!");
    std::vector<const Operator *> Pv, Gv;
    std::vector<const ParOperator *> Av, Xv;
    for (int l = 0; l + 1 < nlevels; l++) Pv.push_back(P[l]->op.get());
    for (int l = 0; l < nlevels; l++) Av.push_back(A[l]->op.get());
    if (G)
      for (int l = 0; l < nlevels; l++) Gv.push_back(G[l] ? G[l]->op.get() : nullptr), Xv.push_back(A_aux[l] ? A_aux[l]->op.get() : nullptr);
    auto *s = new pa_solver;
    s->ctx = ctx;
    // the coarse solver object is kept alive by the multigrid solver; its Solver is borrowed
    struct Borrowed : Solver {
      Solver *inner;
      explicit Borrowed(Solver *i) : inner(i) {}
      void SetOperator(const Operator &) override {}
      void Mult(const Vector &x, Vector &y) const override {
        inner->SetInitialGuess(false);
        inner->Mult(x, y);
      }
      void CheckStatus() const override { inner->CheckStatus(); }
    };
    s->owned.push_back(coarse);
    // a Krylov coarse solve inside the cycle must not stall the stream (see GeometricMultigridSolver's constructor)
    if (auto *cg = dynamic_cast<CgSolver *>(coarse->solver.get())) cg->SetLookahead(-1);
    auto g = std::make_unique<GeometricMultigridSolver>(ctx->ctx, std::make_unique<Borrowed>(coarse->solver.get()),
                                                        Pv, cycle_it, smooth_it, cheby_order, sf_max, sf_min,
                                                        fourth != 0, G ? &Gv : nullptr);
    g->SetOperators(Av, G ? &Xv : nullptr);
    s->solver = std::move(g);
    *S = s;
  }
}
int pa_orthogonalize_column(pa_context *ctx, int kind, int m, const double *const *V, double *w, int n, double *H,
                            pa_par_op *weight) {
  return guarded([&] {
    PA_REQUIRE(ctx && kind >= 0 && kind <= 2 && m >= 0 && (m == 0 || (V && H)) && w && n >= 0, "bad argument");
    std::vector<Vector> basis;
    basis.reserve((size_t)m);
    for (int j = 0; j < m; j++) basis.emplace_back(const_cast<double *>(V[j]), n);
    Vector vw(w, n);
    linalg::OrthogonalizeColumn(ctx->ctx, static_cast<Orthogonalization>(kind), basis, vw, H, m,
                                weight ? weight->op.get() : nullptr);
  });
}
int pa_orthogonalize_column_complex(pa_context *ctx, int kind, int m, const double *const *Vr, const double *const *Vi,
                                    double *wr, double *wi, int n, double *H, pa_par_op *weight) {
  return guarded([&] {
    PA_REQUIRE(ctx && kind >= 0 && kind <= 2 && m >= 0 && (m == 0 || (Vr && Vi && H)) && wr && wi && n >= 0,
               "bad argument");
    std::vector<ComplexVector> basis;
    basis.reserve((size_t)m);
    for (int j = 0; j < m; j++) basis.emplace_back(const_cast<double *>(Vr[j]), const_cast<double *>(Vi[j]), n);
    ComplexVector vw(wr, wi, n);
    std::vector<std::complex<double>> h((size_t)m);
    linalg::OrthogonalizeColumn(ctx->ctx, static_cast<Orthogonalization>(kind), basis, vw, h.data(), m,
                                weight ? weight->op.get() : nullptr);
    for (int j = 0; j < m; j++) H[2 * j] = h[j].real(), H[2 * j + 1] = h[j].imag();
  });
}
int pa_orthonormalize_column(pa_context *ctx, int kind, int m, const double *const *V, double *w, int n, double *H, double *hn) {
  return guarded([&] {
    PA_REQUIRE(ctx && kind >= 0 && kind <= 2 && m >= 0 && (m == 0 || (V && H)) && w && n >= 0 && hn, "bad argument");
    std::vector<Vector> basis;
    basis.reserve((size_t)m);
    for (int j = 0; j < m; j++) basis.emplace_back(const_cast<double *>(V[j]), n);
    Vector vw(w, n);
    *hn = linalg::OrthonormalizeColumn(ctx->ctx, static_cast<Orthogonalization>(kind), basis, vw, H, m);
  });
}
int pa_orthonormalize_column_complex(pa_context *ctx, int kind, int m, const double *const *Vr, const double *const *Vi,
                                     double *wr, double *wi, int n, double *H, double *hn) {
  return guarded([&] {
    PA_REQUIRE(ctx && kind >= 0 && kind <= 2 && m >= 0 && (m == 0 || (Vr && Vi && H)) && wr && wi && n >= 0 && hn,
               "bad argument");
    std::vector<ComplexVector> basis;
    basis.reserve((size_t)m);
    for (int j = 0; j < m; j++) basis.emplace_back(const_cast<double *>(Vr[j]), const_cast<double *>(Vi[j]), n);
    ComplexVector vw(wr, wi, n);
    std::vector<std::complex<double>> h((size_t)m);
    *hn = linalg::OrthonormalizeColumn(ctx->ctx, static_cast<Orthogonalization>(kind), basis, vw, h.data(), m);
    for (int j = 0; j < m; j++) H[2 * j] = h[j].real(), H[2 * j + 1] = h[j].imag();
  });
}
int pa_set_device_orthogonalization(int on) {
  return guarded([&] { linalg::SetDeviceOrthogonalization(on != 0); });
}
long long pa_orthog_resident_columns(void) { return linalg::ResidentColumns(); }
int pa_gmres_set_orthogonalization(pa_solver *S, int kind) {
  return guarded([&] {
    PA_REQUIRE(S && kind >= 0 && kind <= 2, "bad argument");
    auto *g = dynamic_cast<GmresSolver *>(S->solver.get());
    PA_REQUIRE(g, "not a GMRES solver");
    g->SetOrthogonalization(static_cast<Orthogonalization>(kind));
  });
}
int pa_solver_mult(pa_solver *S, const double *b, double *x, int initial_guess) {
  return guarded([&] {
    const int n = S->solver->Height();
    Vector vb(const_cast<double *>(b), n), vx(x, n);
    S->solver->SetInitialGuess(initial_guess != 0);
    S->solver->Mult(vb, vx);
  });
}
struct pa_range {
  PhaseRange r;
  explicit pa_range(const char *name) : r(name) {}
};
pa_range *pa_range_push(const char *name) { return new pa_range(name ? name : ""); }
void pa_range_pop(pa_range *r) { delete r; }
int pa_solver_check_status(const pa_solver *S) {
  return guarded([&] { S->solver->CheckStatus(); });
}
int pa_solver_stats(const pa_solver *S, int *its, double *initial_res, double *final_res, int *converged) {
  return guarded([&] {
    auto *k = dynamic_cast<const IterativeSolver *>(S->solver.get());
    PA_REQUIRE(k, "not an iterative solver");
    if (its) *its = k->GetNumIterations();
    if (initial_res) *initial_res = k->GetInitialRes();
    if (final_res) *final_res = k->GetFinalRes();
    if (converged) *converged = k->GetConverged();
  });
}
void pa_solver_destroy(pa_solver *S) { delete S; }

int pa_complex_op_mult(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, const double *xr, const double *xi, double *yr,
                       double *yi) {
  return guarded([&] {
    PA_REQUIRE(ctx && (Ar || Ai), "null argument");
    ComplexWrapperOperator A(ctx->ctx, Ar ? Ar->op.get() : nullptr, Ai ? Ai->op.get() : nullptr);
    const int n = A.Height();
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), n), y(yr, yi, n);
    A.Mult(x, y);
  });
}
struct pa_complex_op {
  std::unique_ptr<ComplexWrapperOperator> A;
};
int pa_complex_op_create(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, pa_complex_op **A) {
  return guarded([&] {
    PA_REQUIRE(ctx && (Ar || Ai) && A, "null argument");
    auto *p = new pa_complex_op;
    p->A = std::make_unique<ComplexWrapperOperator>(ctx->ctx, Ar ? Ar->op.get() : nullptr, Ai ? Ai->op.get() : nullptr);
    *A = p;
  });
}
int pa_complex_op_apply(pa_complex_op *A, const double *xr, const double *xi, double *yr, double *yi) {
  return guarded([&] {
    PA_REQUIRE(A && xr && xi && yr && yi, "null argument");
    const int n = A->A->Height();
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), n), y(yr, yi, n);
    A->A->Mult(x, y);
  });
}
void pa_complex_op_destroy(pa_complex_op *A) { delete A; }
int pa_complex_gmres_create(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, pa_solver *precond, double rel_tol,
                            double abs_tol, int max_it, int restart, int print, pa_csolver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && (Ar || Ai) && S, "null argument");
    auto *s = new pa_csolver;
    s->ctx = ctx;
    s->A = std::make_unique<ComplexWrapperOperator>(ctx->ctx, Ar ? Ar->op.get() : nullptr, Ai ? Ai->op.get() : nullptr);
    auto g = std::make_unique<ComplexGmresSolver>(ctx->ctx, print);
    g->SetOperator(*s->A);
    if (precond) g->SetPreconditioner(*precond->solver);
    g->SetTol(rel_tol), g->SetAbsTol(abs_tol), g->SetMaxIter(max_it);
    g->SetRestartDim(restart);
    s->solver = std::move(g);
    *S = s;
  });
}
int pa_csolver_mult(pa_csolver *S, const double *br, const double *bi, double *xr, double *xi, int initial_guess) {
  return guarded([&] {
    const int n = S->Height();
    ComplexVector b(const_cast<double *>(br), const_cast<double *>(bi), n), x(xr, xi, n);
    S->solver->Mult(b, x, initial_guess != 0);
  });
}
int pa_csolver_stats(const pa_csolver *S, int *its, double *initial_res, double *final_res, int *converged) {
  return guarded([&] {
    if (its) *its = S->solver->GetNumIterations();
    if (initial_res) *initial_res = S->solver->GetInitialRes();
    if (final_res) *final_res = S->solver->GetFinalRes();
    if (converged) *converged = S->solver->GetConverged();
  });
}
void pa_csolver_destroy(pa_csolver *S) { delete S; }

/* Solver<ComplexOperator> smoothers on a ComplexParOperator: kind 0 JacobiSmoother, 1 ChebyshevSmoother (4th kind),
 * 2 ChebyshevSmoother1stKind (linalg/jacobi.cpp, chebyshev.cpp:160-293) */
struct pa_cprecond {
  pa_context *ctx;
  std::unique_ptr<ComplexSolver> solver;
  int n = 0;
  pa_solver *owned_real = nullptr;  // real coarse solver wrapped by a multigrid solver
  ~pa_cprecond() {
    solver.reset();
    delete owned_real;
  }
};
int pa_complex_smoother_create(pa_context *ctx, pa_complex_par_op *A, int kind, int smooth_it, int order, double sf_max,
                               double sf_min, pa_cprecond **P) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && P, "null argument");
    auto p = std::make_unique<pa_cprecond>();
    p->ctx = ctx, p->n = A->op->Height();
    if (kind == 0)
      p->solver = std::make_unique<ComplexJacobiSmoother>(ctx->ctx);
    else
      p->solver = std::make_unique<ComplexChebyshevSmoother>(ctx->ctx, smooth_it, order, sf_max, kind == 1, sf_min);
    p->solver->SetOperator(*A->op);
    *P = p.release();
  });
}
int pa_complex_smoother_lambda_max(const pa_cprecond *P, double *lambda_max) {
  return guarded([&] {
    auto *c = dynamic_cast<const ComplexChebyshevSmoother *>(P ? P->solver.get() : nullptr);
    PA_REQUIRE(c && lambda_max, "not a Chebyshev smoother");
    *lambda_max = c->LambdaMax();
  });
}
/* y = B x (initial_guess == 0) or y <- y + B (x - A y) */
int pa_complex_smoother_mult(pa_cprecond *P, const double *xr, const double *xi, double *yr, double *yi, int initial_guess) {
  return guarded([&] {
    PA_REQUIRE(P && xr && xi && yr && yi, "null argument");
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), P->n), y(yr, yi, P->n);
    P->solver->SetInitialGuess(initial_guess != 0);
    P->solver->Mult(x, y);
    P->solver->SetInitialGuess(false);
  });
}
int pa_csolver_set_complex_preconditioner(pa_csolver *S, pa_cprecond *P) {
  return guarded([&] {
    PA_REQUIRE(S && P, "null argument");
    S->solver->SetPreconditioner(*P->solver);
  });
}
/* GeometricMultigridSolver<ComplexOperator> (gmg.cpp:16-205) over ComplexParOperators A[0 .. nlevels) (coarsest first), real
 * prolongations P[0 .. nlevels - 1), complex Chebyshev smoothers and `coarse`, a real solver applied to the real and the
 * imaginary part of the coarsest level (MfemWrapperSolver); takes ownership of `coarse` */
int pa_complex_gmg_create(pa_context *ctx, int nlevels, pa_complex_par_op *const *A, pa_interp *const *P,
                          pa_complex_par_op *const *A_aux, pa_interp *const *G, pa_solver *coarse, int cycle_it, int smooth_it,
                          int cheby_order, double sf_max, double sf_min, int fourth, pa_cprecond **out) {
  return guarded([&] {
    PA_REQUIRE(ctx && nlevels >= 1 && A && coarse && out, "Empty finite element space hierarchy during multigrid solver setup!");
    PA_REQUIRE((A_aux == nullptr) == (G == nullptr), "auxiliary operators and discrete gradients come together");
    std::vector<const Operator *> Pv, Gv;
    std::vector<const ComplexParOperator *> Av, Xv;
    for (int l = 0; l + 1 < nlevels; l++) Pv.push_back(P[l]->op.get());
    for (int l = 0; l < nlevels; l++) Av.push_back(A[l]->op.get());
    if (G)
      for (int l = 0; l < nlevels; l++) Gv.push_back(G[l] ? G[l]->op.get() : nullptr), Xv.push_back(A_aux[l] ? A_aux[l]->op.get() : nullptr);
    auto p = std::make_unique<pa_cprecond>();
    p->ctx = ctx, p->n = Av.back()->Height(), p->owned_real = coarse;
    auto g = std::make_unique<ComplexGeometricMultigridSolver>(ctx->ctx, std::make_unique<ComplexWrapperSolver>(*coarse->solver), Pv,
                                                               cycle_it, smooth_it, cheby_order, sf_max, sf_min, fourth != 0,
                                                               G ? &Gv : nullptr);
    g->SetOperators(Av, G ? &Xv : nullptr);
    p->solver = std::move(g);
    *out = p.release();
  });
}
int pa_complex_gmg_smoother_lambda_max(const pa_cprecond *P, int level, double *lambda_max) {
  return guarded([&] {
    auto *g = dynamic_cast<const ComplexGeometricMultigridSolver *>(P ? P->solver.get() : nullptr);
    PA_REQUIRE(g && lambda_max, "not a complex multigrid solver");
    const ComplexSolver &s = g->Smoother(level);
    if (auto *d = dynamic_cast<const ComplexDistRelaxationSmoother *>(&s)) {  // {primary, auxiliary}
      lambda_max[0] = d->Primary().LambdaMax(), lambda_max[1] = d->Auxiliary().LambdaMax();
      return;
    }
    auto *c = dynamic_cast<const ComplexChebyshevSmoother *>(&s);
    PA_REQUIRE(c, "level smoother is not a Chebyshev smoother");
    lambda_max[0] = c->LambdaMax(), lambda_max[1] = 0.0;
  });
}
void pa_complex_smoother_destroy(pa_cprecond *P) { delete P; }

/* ---- ComplexParOperator (linalg/rap.cpp:393-749) over two local operators ---------------------------------------- */
int pa_complex_par_op_create(pa_context *ctx, pa_op *Ar, pa_op *Ai, int n_true, pa_halo *halo, pa_complex_par_op **A) {
  return guarded([&] {
    PA_REQUIRE(ctx && (Ar || Ai) && A, "null argument");
    auto *p = new pa_complex_par_op;
    p->ctx = ctx;
    if (Ar) p->local_r = std::make_unique<ceed::Operator>(ctx->ctx, Ar, false);
    if (Ai) p->local_i = std::make_unique<ceed::Operator>(ctx->ctx, Ai, false);
    p->op = std::make_unique<ComplexParOperator>(ctx->ctx, p->local_r.get(), p->local_i.get(), n_true,
                                                 halo ? halo->halo.get() : nullptr);
    *A = p;
  });
}
int pa_complex_par_op_set_essential(pa_complex_par_op *A, const int32_t *ess, int n_ess, int policy) {
  return guarded([&] {
    PA_REQUIRE(A && (ess || n_ess == 0), "null argument");
    A->op->SetEssentialTrueDofs(ess, n_ess, policy == PA_DIAG_ONE ? ParOperator::DiagonalPolicy::DIAG_ONE
                                                                  : ParOperator::DiagonalPolicy::DIAG_ZERO);
  });
}
/* mode 0: y = A x, 1: y = A^T x, 2: y = A^H x; add != 0: y += (a_re + i a_im) op(A) x */
int pa_complex_par_op_mult(pa_complex_par_op *A, int mode, int add, double a_re, double a_im, const double *xr,
                           const double *xi, double *yr, double *yi) {
  return guarded([&] {
    PA_REQUIRE(A && xr && xi && yr && yi && mode >= 0 && mode <= 2, "bad argument");
    const int n = A->op->Height();
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), n), y(yr, yi, n);
    const std::complex<double> a(a_re, a_im);
    if (!add) {
      if (mode == 0) A->op->Mult(x, y);
      if (mode == 1) A->op->MultTranspose(x, y);
      if (mode == 2) A->op->MultHermitianTranspose(x, y);
    } else {
      if (mode == 0) A->op->AddMult(x, y, a);
      if (mode == 1) A->op->AddMultTranspose(x, y, a);
      if (mode == 2) A->op->AddMultHermitianTranspose(x, y, a);
    }
  });
}
/* The same three forms of the LOCAL ComplexWrapperOperator (linalg/operator.cpp:58-413) on L-vectors. */
int pa_complex_par_op_local_mult(pa_complex_par_op *A, int mode, int add, double a_re, double a_im, const double *xr,
                                 const double *xi, double *yr, double *yi) {
  return guarded([&] {
    PA_REQUIRE(A && xr && xi && yr && yi && mode >= 0 && mode <= 2, "bad argument");
    const ComplexOperator &L = A->op->LocalOperator();
    const int n = L.Height();
    ComplexVector x(const_cast<double *>(xr), const_cast<double *>(xi), n), y(yr, yi, n);
    const std::complex<double> a(a_re, a_im);
    if (!add) {
      if (mode == 0) L.Mult(x, y);
      if (mode == 1) L.MultTranspose(x, y);
      if (mode == 2) L.MultHermitianTranspose(x, y);
    } else {
      if (mode == 0) L.AddMult(x, y, a);
      if (mode == 1) L.AddMultTranspose(x, y, a);
      if (mode == 2) L.AddMultHermitianTranspose(x, y, a);
    }
  });
}
int pa_complex_par_op_assemble_diagonal(pa_complex_par_op *A, double *dr, double *di) {
  return guarded([&] {
    PA_REQUIRE(A && dr && di, "null argument");
    ComplexVector d(dr, di, A->op->Height());
    A->op->AssembleDiagonal(d);
  });
}
void pa_complex_par_op_destroy(pa_complex_par_op *A) { delete A; }

/* GmresSolver / FgmresSolver <ComplexOperator> on a ComplexParOperator: flexible != 0 is FGMRES, pc_side 0 left /
 * 1 right (iterative.hpp:187-272), orthog 0 MGS / 1 CGS / 2 CGS2. */
int pa_complex_gmres_create_par(pa_context *ctx, pa_complex_par_op *A, pa_solver *precond, double rel_tol, double abs_tol,
                                int max_it, int restart, int flexible, int pc_side, int orthog, int print, pa_csolver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && S && orthog >= 0 && orthog <= 2, "bad argument");
    auto *s = new pa_csolver;
    s->ctx = ctx;
    s->Aext = A->op.get();
    auto g = std::make_unique<ComplexGmresSolver>(ctx->ctx, print, flexible != 0);
    g->SetOperator(*s->Aext);
    if (precond) g->SetPreconditioner(*precond->solver);
    g->SetTol(rel_tol), g->SetAbsTol(abs_tol), g->SetMaxIter(max_it);
    g->SetRestartDim(restart);
    g->SetOrthogonalization(static_cast<Orthogonalization>(orthog));
    if (!flexible) g->SetPreconditionerSide(pc_side ? PreconditionerSide::RIGHT : PreconditionerSide::LEFT);
    s->solver = std::move(g);
    *S = s;
  });
}

/* CgSolver<ComplexOperator> (iterative.cpp:360-486) on a ComplexParOperator: Hermitian positive definite systems */
int pa_complex_cg_create_par(pa_context *ctx, pa_complex_par_op *A, pa_solver *precond, double rel_tol, double abs_tol, int max_it,
                             int print, pa_csolver **S) {
  return guarded([&] {
    PA_REQUIRE(ctx && A && S, "bad argument");
    auto *s = new pa_csolver;
    s->ctx = ctx;
    s->Aext = A->op.get();
    s->solver = std::make_unique<ComplexCgSolver>(ctx->ctx, print);
    s->solver->SetOperator(*s->Aext);
    if (precond) s->solver->SetPreconditioner(*precond->solver);
    s->solver->SetTol(rel_tol), s->solver->SetAbsTol(abs_tol), s->solver->SetMaxIter(max_it);
    *S = s;
  });
}

/* GmresSolver::SetPreconditionerSide (iterative.hpp:214): 0 left (default), 1 right */
int pa_gmres_set_pc_side(pa_solver *S, int side) {
  return guarded([&] {
    auto *g = dynamic_cast<GmresSolver *>(S ? S->solver.get() : nullptr);
    PA_REQUIRE(g, "not a GMRES solver");
    g->SetPreconditionerSide(side ? PreconditionerSide::RIGHT : PreconditionerSide::LEFT);
  });
}

/* ParOperator::MultTranspose (rap.cpp:236-275) */
int pa_par_op_mult_transpose(pa_par_op *A, const double *x, double *y) {
  return guarded([&] {
    PA_REQUIRE(A && x && y, "null argument");
    Vector vx(const_cast<double *>(x), A->op->Width()), vy(y, A->op->Height());
    A->op->MultTranspose(vx, vy);
  });
}

int pa_interp_create(pa_context *ctx, const pa_restriction_desc *rc, const pa_basis_desc *bc,
                     const pa_restriction_desc *rf, const pa_basis_desc *bf, const double *Ic, const double *Io,
                     pa_halo *coarse_halo, int nt_c, int nt_f, pa_interp **P) {
  return guarded([&] {
    PA_REQUIRE(ctx && rc && bc && rf && bf && P, "null argument");
    auto *p = new pa_interp;
    p->ctx = ctx;
    p->op.reset(make_interp_operator(ctx->ctx, *rc, *bc, *rf, *bf, Ic, Io,
                                     coarse_halo ? coarse_halo->halo.get() : nullptr, nt_c, nt_f, 0));
    *P = p;
  });
}
int pa_gradient_create(pa_context *ctx, const pa_restriction_desc *rh, const pa_basis_desc *bh,
                       const pa_restriction_desc *rn, const pa_basis_desc *bn, const double *Dg, pa_halo *h1_halo,
                       int nt_h1, int nt_nd, pa_interp **G) {
  return guarded([&] {
    PA_REQUIRE(ctx && rh && bh && rn && bn && Dg && G, "null argument");
    const int n = bn->order + 1;
    std::vector<double> I((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) I[(size_t)i * n + i] = 1.0;
    auto *p = new pa_interp;
    p->ctx = ctx;
    p->op.reset(make_interp_operator(ctx->ctx, *rh, *bh, *rn, *bn, I.data(), Dg,
                                     h1_halo ? h1_halo->halo.get() : nullptr, nt_h1, nt_nd, 1));
    *G = p;
  });
}
int pa_interp_create_dense(pa_context *ctx, const pa_restriction_desc *dom, const pa_restriction_desc *range,
                           const double *M, pa_halo *dom_halo, int nt_dom, int nt_range, pa_interp **P) {
  return guarded([&] {
    PA_REQUIRE(ctx && dom && range && M && P, "null argument");
    auto *p = new pa_interp;
    p->ctx = ctx;
    p->op.reset(make_dense_interp_operator(ctx->ctx, *dom, *range, M, dom_halo ? dom_halo->halo.get() : nullptr, nt_dom,
                                           nt_range));
    *P = p;
  });
}
int pa_interp_create_refinement(pa_context *ctx, const pa_restriction_desc *dom, const pa_restriction_desc *range, int nmat,
                                const double *M, const uint8_t *mat_id, pa_halo *dom_halo, int nt_dom, int nt_range, pa_interp **P) {
  return guarded([&] {
    PA_REQUIRE(ctx && dom && range && M && mat_id && P, "null argument");
    auto *p = new pa_interp;
    p->ctx = ctx;
    p->op.reset(make_dense_interp_operator(ctx->ctx, *dom, *range, M, dom_halo ? dom_halo->halo.get() : nullptr, nt_dom,
                                           nt_range, nmat, mat_id));
    *P = p;
  });
}
int pa_interp_mult(pa_interp *P, const double *x, double *y) {
  return guarded([&] {
    Vector vx(const_cast<double *>(x), P->op->Width()), vy(y, P->op->Height());
    P->op->Mult(vx, vy);
  });
}
int pa_interp_mult_transpose(pa_interp *P, const double *x, double *y) {
  return guarded([&] {
    Vector vx(const_cast<double *>(x), P->op->Height()), vy(y, P->op->Width());
    P->op->MultTranspose(vx, vy);
  });
}
void pa_interp_destroy(pa_interp *P) { delete P; }

}  // extern "C"
