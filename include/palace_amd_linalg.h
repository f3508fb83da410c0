/*
 * palace_amd_linalg.h — C ABI over the C++ host layer that mirrors palace::linalg on HBM vectors
 * (palace_amd/csrc/linalg.hpp).  Palace itself would use the C++ classes directly (same names:
 * ParOperator, CgSolver, GmresSolver, ChebyshevSmoother, GeometricMultigridSolver); this C face
 * exists so the same objects can be driven from tests / bench (ctypes) and from other languages.
 * Same rules as palace_amd.h: 0 = success, pa_last_error() for the message, device pointers for
 * vectors, host pointers for descriptors (copied).
 */
#ifndef PALACE_AMD_LINALG_H
#define PALACE_AMD_LINALG_H

#include "palace_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_context pa_context; /* stream + communicator                          */
typedef struct pa_halo pa_halo;       /* conforming prolongation P of one space (MPI halo in the reference) */
typedef struct pa_par_op pa_par_op;   /* palace::ParOperator, linalg/rap.cpp             */
typedef struct pa_interp pa_interp;   /* p-prolongation / discrete gradient, fem/bilinearform.cpp:203-282 */
typedef struct pa_solver pa_solver;   /* palace::Solver<Operator>, linalg/solver.hpp     */
typedef struct pa_csolver pa_csolver; /* Krylov solver on ComplexOperator (ComplexVector = two real vectors) */
typedef struct pa_complex_par_op pa_complex_par_op; /* palace::ComplexParOperator, linalg/rap.hpp:124-221 */

/* --- context: everything created from it is enqueued on `stream` (a hipStream_t) -------------
 * stream == NULL: the context creates a stream of its own with the default (blocking) flag, which is ordered against
 * the legacy null stream exactly like the null stream itself (callers that fill vectors on the null stream keep their
 * ordering) and which, unlike the null stream, can be recorded into HIP graphs. */
int pa_context_create(void *stream, pa_context **ctx);
/* the hipStream_t the context enqueues on */
int pa_context_stream(const pa_context *ctx, void **stream);
void pa_context_destroy(pa_context *ctx);
int pa_context_synchronize(pa_context *ctx);

/* --- RCCL communicator (replaces MPI_Comm on the hot path: utils/communication.hpp:249-273) -- */
/* rank 0 fills 128 bytes; the caller broadcasts them to all ranks by any means */
int pa_comm_unique_id(char *out128);
int pa_context_init_comm(pa_context *ctx, int rank, int size, const char *unique_id128);
/* Test harness for the multi-rank code paths on a one-GPU machine: `size` ranks as threads of ONE process, each with its own
 * context (and stream); collectives rendezvous on a host barrier and copy through device memory instead of RCCL.  Every rank
 * thread must make the same sequence of collective calls.  Not a product path (tests/test_multirank_local_gpu.py). */
typedef struct pa_local_group pa_local_group;
int pa_local_group_create(int size, pa_local_group **group);
void pa_local_group_destroy(pa_local_group *group);
/* a rank thread that fails calls this so that the others leave their barriers with an error instead of waiting for ever */
void pa_local_group_abort(pa_local_group *group);
int pa_context_init_comm_local(pa_context *ctx, int rank, pa_local_group *group);
/* Peer transport (palace_amd/csrc/comm.hpp): halo exchanges and global sums as direct stores into the other ranks' device
 * memory (xGMI between the GPUs of a node) -- plain kernels on the context's stream, recordable in HIP graphs.  Every rank owns
 * one arena; the caller gathers the 64-byte IPC handles of all ranks (MPI_Allgather in Palace: utils/communication.hpp) and
 * hands the table to pa_comm_peer_connect.  pa_context_init_comm_peer makes a communicator that uses nothing else (no RCCL:
 * also what lets two processes share one GPU); on an RCCL communicator (pa_context_init_comm) connecting the arenas moves the
 * halo exchanges and the sums to the peer transport as well (PALACE_AMD_HALO=rccl keeps RCCL's send / receive groups).
 * pa_comm_peer_check raises if a wait of the transport has timed out (a rank died, mismatched plans). */
int pa_context_init_comm_peer(pa_context *ctx, int rank, int size);
int pa_comm_peer_handle(pa_context *ctx, char *out64);
int pa_comm_peer_connect(pa_context *ctx, const char *handles /* [size][64] */);
int pa_comm_peer_disconnect(pa_context *ctx); /* RCCL communicators: back to send / receive groups and ncclAllReduce */
int pa_comm_peer_ready(const pa_context *ctx);
int pa_comm_peer_check(pa_context *ctx);
/* Ordering tier of the transport (process-wide): 0 (default) relaxed system-scope atomics + s_waitcnt around the flag stores,
 * 1 system-scope release / acquire fences (an L2 write-back per exchange kernel: slower; the fall-back if the relaxed protocol
 * fails pa_comm_peer_stress on a machine).  PALACE_AMD_PEER_FENCE=1 selects it from the environment. */
int pa_comm_peer_set_fenced(int on);
int pa_comm_peer_fenced(void);
/* time limit of the device-side waits of the transport, seconds (default 60, PALACE_AMD_PEER_TIMEOUT_S): after it a wait gives
 * up, the rank's error word is raised and every host synchronisation point of the library (dots, sums, solver statistics,
 * pa_context_synchronize, pa_comm_peer_check) returns an error.  Longer than any skew between ranks. */
int pa_comm_peer_set_timeout(double seconds);
/* Stress test: `rounds` rounds of P, P^T (direct = 0: through the local vector, 1: the direct form of ParOperator::Mult, ghosts
 * read from the mailbox in place) and a global sum on a RING plan (`ring`: pa_halo over a vector of 2 n entries -- owned
 * [0, n) sent to rank + 1, ghosts [n, 2 n) owned by rank - 1), payloads that change every round, every received value checked
 * on the device against its closed form; graph != 0: the round is recorded once and replayed (what the solvers do).
 * *failures = wrong values seen by this rank; a timed-out wait is an error return.  Collective. */
int pa_comm_peer_stress(pa_context *ctx, pa_halo *ring, int n, int rounds, int direct, int graph, long long *failures);
int pa_halo_uses_peer(const pa_halo *halo);
int pa_context_rank(const pa_context *ctx);
int pa_context_size(const pa_context *ctx);
/* in-place sum of n doubles (device memory) over all ranks */
int pa_allreduce_sum(pa_context *ctx, double *dev_buf, int n);

/* Halo plan of one space.  Local vector layout: true (owned) dofs [0, n_true), then ghosts.
 *   nbr[k]                         neighbour rank
 *   send_idx[send_off[k]..send_off[k+1])  owned local dofs whose values neighbour k needs
 *   recv_idx[recv_off[k]..recv_off[k+1])  ghost slots owned by neighbour k
 * Both sides must list a pair's dofs in the same order. */
int pa_halo_create(pa_context *ctx, int nnbr, const int *nbr, const int *send_off,
                   const int32_t *send_idx, const int *recv_off, const int32_t *recv_idx,
                   pa_halo **halo);
void pa_halo_destroy(pa_halo *halo);
/* P and P^T on a local vector (device pointer) — exposed for tests */
int pa_halo_prolongate(pa_context *ctx, pa_halo *halo, double *lx);
int pa_halo_restrict_add(pa_context *ctx, pa_halo *halo, double *ly);

/* --- ParOperator (linalg/rap.cpp:154-234) ---------------------------------------------------- */
enum pa_diag_policy { PA_DIAG_ZERO = 0, PA_DIAG_ONE = 1 };
/* `local` stays owned by the caller and must outlive the ParOperator; halo may be NULL (one rank) */
int pa_par_op_create(pa_context *ctx, pa_op *local, int n_true, const int32_t *ess_tdofs, int n_ess,
                     int diag_policy, pa_halo *halo, pa_par_op **A);
/* BuildParSumOperator (linalg/rap.cpp:843-919): ParOperator around sum_k coeffs[k] * locals[k] (the
 * BaseSumOperator of linalg/operator.hpp:132-270), e.g. a0 K + a1 C + a2 M.  The locals stay owned by the caller. */
int pa_par_sum_op_create(pa_context *ctx, int nterms, pa_op *const *locals, const double *coeffs, int n_true,
                         const int32_t *ess_tdofs, int n_ess, int diag_policy, pa_halo *halo, pa_par_op **A);
/* ParOperator around an assembled local operator (ParOperator::ParallelAssemble, linalg/rap.cpp:84-152: the reference
 * assembles the coarsest multigrid level because its solver needs a matrix).  `csr` comes from pa_op_full_assemble,
 * stays owned by the caller and must outlive the ParOperator; the local apply is a device CSR matrix-vector product. */
int pa_par_op_create_assembled(pa_context *ctx, const pa_csr *csr, int n_true, const int32_t *ess_tdofs, int n_ess,
                               int diag_policy, pa_halo *halo, pa_par_op **A);
void pa_par_op_destroy(pa_par_op *A);
int pa_par_op_mult(pa_par_op *A, const double *x, double *y);
/* The direct form of a multi-rank Mult (peer transport + pa_op_mult_split: no L-vector copies).
 * pa_par_op_direct_form: 1 in use, 0 available but switched off, -1 not available for this operator / transport.
 * pa_par_op_set_direct(A, 0) selects the L-vector form (A / B runs, cross-checks of the two forms); (A, 1) back. */
int pa_par_op_direct_form(const pa_par_op *A);
int pa_par_op_set_direct(pa_par_op *A, int on);
/* ParOperator::AddMult (rap.cpp:277-318): y += a (P^T A P with the essential-dof handling) x. */
int pa_par_op_add_mult(pa_par_op *A, const double *x, double *y, double a);
/* ParOperator::EliminateRHS (rap.cpp:56-82): b -= A_unconstrained x|ess ; b[ess] = x[ess] (DIAG_ONE) or 0. */
int pa_par_op_eliminate_rhs(pa_par_op *A, const double *x, double *b);
int pa_par_op_assemble_diagonal(pa_par_op *A, double *diag);

/* --- vectors (linalg/vector.cpp) ------------------------------------------------------------- */
int pa_vec_dot(pa_context *ctx, const double *x, const double *y, int n, double *result);
/* linalg::Sum (global sum of the entries; LocalSum vector.cpp:687-699 + Mpi::GlobalSum; a ComplexVector's sum is the
 * pair of its parts' sums) and linalg::Sqrt (x = sqrt(s x), vector.cpp:774-781); unit tests test/unit/test-vector.cpp. */
int pa_vec_sum(pa_context *ctx, const double *x, int n, double *result);
int pa_vec_sqrt(pa_context *ctx, double *x, int n, double s);
int pa_vec_axpby(pa_context *ctx, double a, const double *x, double b, double *y, int n);
/* Measurement aid (SURVEY.md 8d "measure both peaks in the same run"): launches n_blocks x 256 threads that each
 * issue `iters` rounds of eight independent v_mfma_f64_16x16x4_f64; *flops_per_launch receives the flop count.
 * Time it with events on the context's stream; `scratch` is any device buffer of >= 1 double (never written). */
int pa_bench_mfma_f64(pa_context *ctx, int iters, int n_blocks, double *scratch, double *flops_per_launch);
int pa_vec_set_random(pa_context *ctx, double *x, int n, uint64_t seed);
/* ComplexVector members (linalg/vector.hpp:95-146, vector.cpp:172-460) on split real / imaginary device arrays.
 * op: 0 x *= a | 1 x = |x| | 2 x = 1 ./ x | 3 x = conj(x) | 4 y = a x + b y | 5 z = a x + b y + c z |
 * 6 out[0..1] = x^T y (no conjugate; global).  coef: the complex coefficients a, b, c as (re, im) pairs (NULL: ones). */
int pa_cvec_op(pa_context *ctx, int op, int n, const double *coef, double *xr, double *xi, double *yr, double *yi, double *zr,
               double *zi, double *out);
/* ComplexVector::SetBlocks: x = [s_0 y_0; s_1 y_1; ...] (s: (re, im) pairs, NULL = ones) */
int pa_cvec_set_blocks(pa_context *ctx, double *xr, double *xi, int n, int nblocks, const double *const *yr,
                       const double *const *yi, const int *sizes, const double *s);
/* DiagonalOperator / ComplexDiagonalOperator (linalg/operator.hpp:354-423): y (+)= a op(diag(d)) x; mode 0 N, 1 T, 2 H;
 * di == xi == yi == NULL selects the real operator */
int pa_diag_op_apply(pa_context *ctx, int n, const double *dr, const double *di, const double *xr, const double *xi, double *yr,
                     double *yi, double ar, double ai, int mode, int add);

/* --- solvers --------------------------------------------------------------------------------- */
/* ChebyshevSmoother / ChebyshevSmoother1stKind (linalg/chebyshev.cpp); SetOperator(A) is done here:
 * diagonal assembly + power iteration for lambda_max (linalg/operator.cpp:583-631). */
int pa_chebyshev_create(pa_context *ctx, pa_par_op *A, int smooth_it, int order, double sf_max,
                        int fourth_kind, pa_solver **S);
int pa_chebyshev_lambda_max(const pa_solver *S, double *lambda_max);
/* *fused = 1 when the smoother's steps run inside the operator's E^T (pa_op_mult_cheb_step, palace_amd.h) instead of as an apply
 * followed by a vector kernel.  level < 0: S is a Chebyshev smoother; level >= 1: that level's smoother of a multigrid solver. */
int pa_chebyshev_fused_step(const pa_solver *S, int level, int *fused);
/* ProductOperator / ComplexProductOperator (linalg/operator.hpp:270-352): y (+)= a op(A B) x; mode 0 N, 1 T, 2 H */
int pa_product_op_apply(pa_par_op *A, pa_par_op *B, const double *x, double *y, int transpose, double a, int add);
int pa_complex_product_op_apply(pa_complex_par_op *A, pa_complex_par_op *B, const double *xr, const double *xi, double *yr,
                                double *yi, int mode, double ar, double ai, int add);
/* CgSolver execution mode.  By default the scalars of the recurrence stay on the device (no host round trip per
 * iteration; an iteration is replayed as a HIP graph) and the host enqueues `lookahead` iterations beyond the last
 * residual it has read; iterates and iteration counts equal those of the synchronous loop of iterative.cpp:360-486,
 * which host_scalars != 0 selects literally.  lookahead < 0: never wait (statistics are read when asked for). */
int pa_cg_set_lookahead(pa_solver *S, int lookahead, int host_scalars);
/* The same for the Chebyshev smoother of level l >= 1 of a GeometricMultigridSolver (diagnostics / parity checks). */
int pa_gmg_smoother_lambda_max(const pa_solver *S, int level, double *lambda_max);
/* ChebyshevSmoother1stKind (linalg/chebyshev.cpp:222-293); sf_min <= 0: the optimised lambda_min estimate (:244-247). */
int pa_chebyshev_create_1st_kind(pa_context *ctx, pa_par_op *A, int smooth_it, int order, double sf_max, double sf_min,
                                 pa_solver **S);
/* DistRelaxationSmoother (linalg/distrelaxation.cpp:14-151) on its own: A the Nedelec ParOperator, A_aux the auxiliary
 * H1 ParOperator, G their discrete gradient; the two Chebyshev smoothers' eigenvalue estimates for diagnostics. */
int pa_dist_relaxation_create(pa_context *ctx, pa_par_op *A, pa_par_op *A_aux, pa_interp *G, int smooth_it,
                              int cheby_smooth_it, int cheby_order, double cheby_sf_max, double cheby_sf_min,
                              int cheby_4th_kind, pa_solver **S);
int pa_dist_relaxation_lambda_max(const pa_solver *S, double *lambda_max, double *lambda_max_aux);
/* Solver::Mult2 / MultTranspose2 (linalg/solver.hpp, the V-cycle's entry points gmg.cpp:184,204):
 * y <- y + B (x - A y), starting from y when initial_guess != 0 and from zero otherwise. */
int pa_solver_mult2(pa_solver *S, const double *x, double *y, int transpose, int initial_guess);
/* JacobiSmoother (linalg/jacobi.cpp) */
int pa_jacobi_create(pa_context *ctx, pa_par_op *A, pa_solver **S);

/* A small global problem solved redundantly by every rank (the coarsest multigrid level across ranks, where the reference runs
 * HYPRE's distributed AMS / BoomerAMG: linalg/ksp.cpp:143-157).  `inner` solves the GLOBAL problem (n_global unknowns, global
 * numbering, identical on every rank -- pa_ams_create / pa_amg_create on the globally assembled matrix); `gather` is a halo plan
 * on a global-numbered vector (send lists: this rank's true dofs by global number, receive lists: the other ranks'); mine[i] is
 * the global number of true dof i and sign[i] = +-1 its orientation relative to the global dof (NULL: all +1; a rank-local mesh may
 * orient an edge against the global numbering).  y = S x on T-vectors: scatter to global positions, gather from all ranks, inner
 * solve, keep the own entries.  Neither `gather` nor `inner` is owned. */
int pa_replicated_solver_create(pa_context *ctx, pa_halo *gather, pa_solver *inner, const int32_t *mine, const double *sign,
                                int n_true, int n_global, pa_solver **S);

/* Native coarse-level solvers (palace_amd/csrc/amg_solver.hpp), standing where the reference calls HYPRE on its coarsest
 * multigrid level: BoomerAmgSolver (linalg/amg.cpp:12-49; wiring linalg/ksp.cpp:187-200) and HypreAmsSolver
 * (linalg/ams.cpp:18-224; ksp.cpp:166-186).  The matrix is the assembled level (pa_op_full_assemble: what
 * ParOperator::ParallelAssemble gives HYPRE, rap.cpp:84-152); `ess` are the essential true dofs ParOperator eliminates in it
 * (rap.cpp:131-149).  Zero / negative option fields take the defaults of AmgOptions / AmsOptions.  The matrix is the COMPLETE
 * problem: the cycles are rank-local (no halo, no global reductions); across ranks wrap the solver of the globally assembled
 * matrix in pa_replicated_solver_create. */
typedef struct {
  int max_levels, coarse_size, smooth_order;
  double theta;
} pa_amg_options;
typedef struct {
  int cycle_it, smooth_order, singular;
  pa_amg_options amg;
} pa_ams_options;
int pa_amg_create(pa_context *ctx, const pa_csr *A, const int32_t *ess, int n_ess, const pa_amg_options *opt, pa_solver **S);
/* G: the discrete gradient [rows of A x n_vert] as host CSR arrays (HYPRE_AMSSetDiscreteGradient, ams.cpp:204), coords: the
 * vertex coordinates [n_vert][dim] on the host (HYPRE_AMSSetCoordinateVectors, ams.cpp:206-210: lowest-order spaces). */
int pa_ams_create(pa_context *ctx, const pa_csr *A, const int32_t *ess, int n_ess, int n_vert, const int32_t *G_rowptr,
                  const int32_t *G_col, const double *G_val, const double *coords, int dim, const pa_ams_options *opt,
                  pa_solver **S);
/* The coarsest level of a MULTI-RANK hierarchy solved redundantly by every rank with the native cycles (the reference runs HYPRE's
 * distributed AMS / BoomerAMG there: linalg/ksp.cpp:129-239).  Built from what each rank holds -- `level0`: the level's ParOperator
 * (halo plan, essential dofs, assembled or partially assembled local operator); G (AMS; NULL: the AMG cycle): the lowest-order
 * discrete gradient between the ranks' H1 and H(curl) true dofs; xyz_true [nv_true][dim]: coordinates of this rank's true
 * vertices -- by numbering the true dofs rank by rank, gathering triplets, gradient rows and coordinates over the communicator and
 * assembling the same global problem everywhere (ksp.hpp: ReplicatedCoarseSolver).  Collective. */
int pa_replicated_coarse_create(pa_context *ctx, pa_par_op *level0, pa_interp *G, int nv_true, const double *xyz_true, int dim,
                                int cycle_it, int singular, pa_solver **S);
/* Round 5: the solver keeps and applies each rank's ROWS of every level of its algebraic hierarchies (amg_dist.hpp; one owner ->
 * ghost exchange per product; where the reference runs HYPRE on the distributed matrix, linalg/ksp.cpp:129-239) unless
 * PALACE_AMD_COARSE_SOLVE=replicated was set when it was created.  *distributed = 1 | 0; *levels (may be NULL): levels of the AMG
 * hierarchy (AMS: of its nodal-space hierarchy) as this rank holds them, 0 for the replicated form. */
int pa_replicated_coarse_info(const pa_solver *S, int *distributed, int *levels);
/* The hierarchy of an AMG solver (which = 0) or of the gradient-space (1) / nodal-space (2) solver inside an AMS solver:
 * number of levels; and copies of its matrices -- kind 0: A_level, 1: P_level (level + 1 -> level), 2: the dense inverse used
 * on the last level (row-major in val, rowptr / col untouched).  Null output arrays: sizes only. */
int pa_amg_num_levels(const pa_solver *S, int which, int *nlevels);
int pa_amg_get_matrix(const pa_solver *S, int which, int level, int kind, int32_t *nrows, int32_t *ncols, int64_t *nnz,
                      int32_t *rowptr, int32_t *col, double *val);
/* CgSolver / GmresSolver / FgmresSolver (linalg/iterative.cpp).  precond may be NULL. */
int pa_cg_create(pa_context *ctx, pa_par_op *A, pa_solver *precond, double rel_tol, double abs_tol,
                 int max_it, int print, pa_solver **S);
int pa_gmres_create(pa_context *ctx, pa_par_op *A, pa_solver *precond, double rel_tol, double abs_tol,
                    int max_it, int restart, int flexible, int print, pa_solver **S);
/* GeometricMultigridSolver (linalg/gmg.cpp): levels 0 (coarsest) .. nlevels-1; P[l] maps level l to
 * l+1; `coarse` solves on level 0 (ownership of `coarse` passes to the multigrid solver). */
int pa_gmg_create(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P,
                  pa_solver *coarse, int cycle_it, int smooth_it, int cheby_order, double cheby_sf_max,
                  double cheby_sf_min, int cheby_4th_kind, pa_solver **S);
/* The same with the Hiptmair (distributive relaxation) smoother on levels >= 1 (linalg/gmg.cpp:41-60,
 * linalg/distrelaxation.cpp): A_aux[l] = auxiliary H1 operator of level l, G[l] = discrete gradient of
 * level l (entries for level 0 may be NULL). */
int pa_gmg_create_aux(pa_context *ctx, int nlevels, pa_par_op *const *A, pa_interp *const *P,
                      pa_par_op *const *A_aux, pa_interp *const *G, pa_solver *coarse, int cycle_it,
                      int smooth_it, int cheby_order, double cheby_sf_max, double cheby_sf_min,
                      int cheby_4th_kind, pa_solver **S);
/* x = S(b) on T-vectors; initial_guess != 0 uses x as the starting iterate (Krylov solvers) */
/* Gram-Schmidt variant of (F)GMRES: 0 = MGS (default), 1 = CGS, 2 = CGS2 (config "Orthogonalization",
 * linalg/orthog.hpp:41-89).  The classical variants do all inner products of a column in one pass over
 * the new vector and one all-reduce. */
/* OrthogonalizeColumnMGS / OrthogonalizeColumnCGS (linalg/orthog.hpp:41-89) as free functions, the form the reference's
 * own unit tests exercise (test/unit/test-orthog.cpp): H[j] = (w, V[j]), w -= sum_j H[j] V[j] for j < m; kind 0 = MGS,
 * 1 = CGS, 2 = CGS with one refinement pass.  V[j], w: device vectors of length n (assumed normalised; w is not
 * normalised on return); H: host.  `weight` (or NULL): (w, v) = v^T W w with a square real operator.  The complex form
 * takes separate real / imaginary arrays (ComplexVector) and returns H as (re, im) pairs, H[j] = V[j]^H (W) w. */
int pa_orthogonalize_column(pa_context *ctx, int kind, int m, const double *const *V, double *w, int n, double *H,
                            pa_par_op *weight);
int pa_orthogonalize_column_complex(pa_context *ctx, int kind, int m, const double *const *Vr, const double *const *Vi,
                                    double *wr, double *wi, int n, double *H, pa_par_op *weight);
/* One Arnoldi column as (F)GMRES does it (linalg/iterative.cpp:629-633, :820-824): orthogonalise w against V[0 .. m) with the
 * chosen Gram-Schmidt variant, *hn = ||w||, w /= *hn.  The coefficients stay on the device between the kernels and the
 * update kernels read them from there: one host synchronisation per column (palace_amd/csrc/orthog.hip).  H as above. */
int pa_orthonormalize_column(pa_context *ctx, int kind, int m, const double *const *V, double *w, int n, double *H, double *hn);
int pa_orthonormalize_column_complex(pa_context *ctx, int kind, int m, const double *const *Vr, const double *const *Vi,
                                     double *wr, double *wi, int n, double *H, double *hn);
/* A / B switch of the above inside (F)GMRES and pa_orthogonalize_column: 0 = the host drives every inner product and update
 * (one synchronisation per inner product, the form of rounds 1-4), 1 = device-resident coefficients (default). */
int pa_set_device_orthogonalization(int on);
/* Round 6: modified Gram-Schmidt columns (one rank, 16-byte aligned vectors that fit the register file: a few million entries)
 * keep w in registers for the whole column -- one pass over every basis vector instead of four vector passes per basis vector
 * (orthog.hip: k_mgs_resident; PALACE_AMD_GS_RESIDENT=0 keeps the chained form).  Returns how many columns this process has
 * orthogonalised that way so far (tests and the bench line tell the two forms apart with it). */
long long pa_orthog_resident_columns(void);
int pa_gmres_set_orthogonalization(pa_solver *S, int kind);
int pa_solver_mult(pa_solver *S, const double *b, double *x, int initial_guess);
/* Named host ranges for profilers (roctx; rocprofv3 --marker-trace).  The library itself brackets the reference's BlockTimer
 * phases (utils/timer.hpp:29-84: "Linear Solve", "Linear Solve / Setup", "Preconditioner", "Coarse Solve", "Operator
 * Construction", "Estimation" ...); a driver adds its own (Timer::TS, EPS, POSTPRO ...) with this pair.  Nestable; no-ops when
 * no roctx library is found or PALACE_AMD_ROCTX=0. */
typedef struct pa_range pa_range;
pa_range *pa_range_push(const char *name);
void pa_range_pop(pa_range *range);

/* Surface what nested solvers could only note on the device while they ran inside a recorded sequence (the coarse PCG of a
 * multigrid cycle meeting a non-finite (Br, r) / (Ap, p): the reference's CheckDot abort, iterative.cpp:39-45).  Krylov
 * solvers call this on their preconditioner after every solve; a caller applying a preconditioner directly may call it
 * whenever it is willing to wait for the stream.  Returns PA_ERROR with the message of the failure, PA_OK otherwise. */
int pa_solver_check_status(const pa_solver *S);
int pa_solver_stats(const pa_solver *S, int *iterations, double *initial_res, double *final_res,
                    int *converged);
void pa_solver_destroy(pa_solver *S);

/* --- complex operators: ComplexWrapperOperator (linalg/operator.cpp:58-134) over two real
 *     ParOperators (real part with its DIAG policy, imaginary part DIAG_ZERO, rap.cpp:450-457) and
 *     the complex instantiation of GmresSolver (linalg/iterative.cpp:543-705) with a real
 *     preconditioner applied to the real and imaginary parts (linalg/gmg.cpp:147-168). ------------ */
int pa_complex_op_mult(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, const double *xr, const double *xi,
                       double *yr, double *yi);
/* Persistent ComplexWrapperOperator (keeps its work vectors): y = (Ar + i Ai) x. */
typedef struct pa_complex_op pa_complex_op;
int pa_complex_op_create(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, pa_complex_op **A);
int pa_complex_op_apply(pa_complex_op *A, const double *xr, const double *xi, double *yr, double *yi);
void pa_complex_op_destroy(pa_complex_op *A);
int pa_complex_gmres_create(pa_context *ctx, pa_par_op *Ar, pa_par_op *Ai, pa_solver *precond, double rel_tol,
                            double abs_tol, int max_it, int restart, int print, pa_csolver **S);
int pa_csolver_mult(pa_csolver *S, const double *br, const double *bi, double *xr, double *xi, int initial_guess);
int pa_csolver_stats(const pa_csolver *S, int *iterations, double *initial_res, double *final_res, int *converged);
void pa_csolver_destroy(pa_csolver *S);
/* Solver<ComplexOperator> smoothers on a ComplexParOperator (the reference's complex-valued preconditioner route):
 * kind 0 JacobiSmoother (linalg/jacobi.cpp), 1 ChebyshevSmoother 4th kind, 2 ChebyshevSmoother1stKind
 * (linalg/chebyshev.cpp:160-293) with the complex inverse diagonal; y = B x or, with initial_guess, y <- y + B (x - A y) */
typedef struct pa_cprecond pa_cprecond;
int pa_complex_smoother_create(pa_context *ctx, pa_complex_par_op *A, int kind, int smooth_it, int order, double sf_max,
                               double sf_min, pa_cprecond **P);
int pa_complex_smoother_lambda_max(const pa_cprecond *P, double *lambda_max);
int pa_complex_smoother_mult(pa_cprecond *P, const double *xr, const double *xi, double *yr, double *yi, int initial_guess);
int pa_csolver_set_complex_preconditioner(pa_csolver *S, pa_cprecond *P);
/* GeometricMultigridSolver<ComplexOperator> (linalg/gmg.cpp:16-205): complex operators and Chebyshev smoothers on every level
 * (coarsest first), the real prolongations on both parts, `coarse` a real solver applied to the real and the imaginary part of
 * level 0 (MfemWrapperSolver, linalg/solver.hpp:67-120; its SetOperator receives the real part of A[0]); takes ownership of it.
 * A_aux / G (both or neither): complex auxiliary-space operators and discrete gradients per level => the smoothers are
 * DistRelaxationSmoother<ComplexOperator> (linalg/distrelaxation.cpp:14-151).  lambda_max: two values per level {primary,
 * auxiliary (0 for plain Chebyshev)} */
int pa_complex_gmg_create(pa_context *ctx, int nlevels, pa_complex_par_op *const *A, pa_interp *const *P,
                          pa_complex_par_op *const *A_aux, pa_interp *const *G, pa_solver *coarse, int cycle_it, int smooth_it,
                          int cheby_order, double sf_max, double sf_min, int fourth, pa_cprecond **out);
int pa_complex_gmg_smoother_lambda_max(const pa_cprecond *P, int level, double *lambda_max);
void pa_complex_smoother_destroy(pa_cprecond *P);

/* --- ComplexParOperator (linalg/rap.hpp:124-221, rap.cpp:393-749): y = P^T (Ar + i Ai) P x over two LOCAL operators
 *     (either may be NULL), essential dofs handled once on the complex vector (rap.cpp:436-462: the real part carries
 *     `policy`, the imaginary part DIAG_ZERO).  mode 0 / 1 / 2 = A / A^T / A^H (Mult, MultTranspose,
 *     MultHermitianTranspose); add != 0 is the AddMult* form y += a op(A) x with complex a.  `local_mult` applies the
 *     local ComplexWrapperOperator (linalg/operator.cpp:58-413) on L-vectors. */
int pa_complex_par_op_create(pa_context *ctx, pa_op *Ar, pa_op *Ai, int n_true, pa_halo *halo, pa_complex_par_op **A);
int pa_complex_par_op_set_essential(pa_complex_par_op *A, const int32_t *ess, int n_ess, int policy);
int pa_complex_par_op_mult(pa_complex_par_op *A, int mode, int add, double a_re, double a_im, const double *xr,
                           const double *xi, double *yr, double *yi);
int pa_complex_par_op_local_mult(pa_complex_par_op *A, int mode, int add, double a_re, double a_im, const double *xr,
                                 const double *xi, double *yr, double *yi);
int pa_complex_par_op_assemble_diagonal(pa_complex_par_op *A, double *dr, double *di);
void pa_complex_par_op_destroy(pa_complex_par_op *A);
/* GmresSolver / FgmresSolver <ComplexOperator> (linalg/iterative.cpp:543-871) on a ComplexParOperator: flexible != 0
 * = FGMRES; pc_side 0 left / 1 right (iterative.hpp:187-272); orthog 0 MGS / 1 CGS / 2 CGS2. */
int pa_complex_gmres_create_par(pa_context *ctx, pa_complex_par_op *A, pa_solver *precond, double rel_tol,
                                double abs_tol, int max_it, int restart, int flexible, int pc_side, int orthog,
                                int print, pa_csolver **S);
/* CgSolver<ComplexOperator> (linalg/iterative.cpp:360-486) on a ComplexParOperator: PCG for Hermitian positive definite systems
 * (complex scalars, inner products y^H x, residual in the preconditioner's inner product); precond: a real solver applied to both
 * parts or NULL.  Solved with pa_csolver_mult / read with pa_csolver_stats like the GMRES solvers. */
int pa_complex_cg_create_par(pa_context *ctx, pa_complex_par_op *A, pa_solver *precond, double rel_tol, double abs_tol, int max_it,
                             int print, pa_csolver **S);
/* GmresSolver::SetPreconditionerSide (iterative.hpp:214): 0 left (default), 1 right. */
int pa_gmres_set_pc_side(pa_solver *S, int side);
/* ParOperator::MultTranspose (linalg/rap.cpp:236-275). */
int pa_par_op_mult_transpose(pa_par_op *A, const double *x, double *y);

/* --- p-prolongation between two spaces on the same mesh (fem/bilinearform.cpp:203-282,
 *     fem/libceed/basis.cpp:116-165 `InitMfemInterpolatorBasis`, fem/libceed/integrator.cpp:515-548):
 *     element-local interpolation applied through E / E^T, output scaled by the inverse (local) dof
 *     multiplicity (bilinearform.cpp:256-279).  The element matrix MFEM's GetTransferMatrix gives
 *     for these nodal tensor elements is the Kronecker product of two 1-D nodal interpolation
 *     matrices, which is what is passed:
 *       Ic [p_f+1][p_c+1]  coarse closed basis evaluated at the fine closed nodes
 *       Io [p_f][p_c]      coarse open basis evaluated at the fine open nodes (HCURL only)
 *     In parallel the wrapper is ParOperator(P, coarse, fine, use_R = true) (fem/fespace.cpp): the
 *     coarse input is halo-prolongated, the fine output restricted to owned dofs. */
int pa_interp_create(pa_context *ctx, const pa_restriction_desc *coarse_restr,
                     const pa_basis_desc *coarse_basis, const pa_restriction_desc *fine_restr,
                     const pa_basis_desc *fine_basis, const double *Ic, const double *Io,
                     pa_halo *coarse_halo, int n_true_coarse, int n_true_fine, pa_interp **P);
/* Discrete gradient G : H1(p) -> ND(p) on the same mesh (the auxiliary-space transfer of the Hiptmair
 * smoother; fem/bilinearform.cpp:203-282 with the gradient interpolator, basis.cpp:139-143).
 *   Dg [p][p+1]  derivative of the closed (Gauss-Lobatto) basis at the open (Gauss-Legendre) nodes
 * Mult = G, MultTranspose = G^T through pa_interp_mult / pa_interp_mult_transpose. */
int pa_gradient_create(pa_context *ctx, const pa_restriction_desc *h1_restr, const pa_basis_desc *h1_basis,
                       const pa_restriction_desc *nd_restr, const pa_basis_desc *nd_basis, const double *Dg,
                       pa_halo *h1_halo, int n_true_h1, int n_true_nd, pa_interp **G);
/* The same two operators for non-tensor elements (tetrahedra, ...): the element projection matrix
 * M [P_range][P_domain] (row-major) MFEM's GetTransferMatrix / discrete-gradient interpolator gives
 * (basis.cpp:132-150), with the native restrictions of both spaces.  When the range space has a dof
 * transformation its restriction must be the one Palace builds for an interpolator range
 * (InvTransformDual, restriction.cpp:318-336). */
int pa_interp_create_dense(pa_context *ctx, const pa_restriction_desc *domain_restr,
                           const pa_restriction_desc *range_restr, const double *M, pa_halo *domain_halo,
                           int n_true_domain, int n_true_range, pa_interp **P);
/* Transfer between the spaces of one finite element collection on a mesh and on its uniform refinement -- the
 * mfem::TransferOperator the reference wraps for two levels on DIFFERENT meshes (fem/fespace.cpp:246-251; the h-levels of
 * ConstructFiniteElementSpaceHierarchy, fem/multigrid.hpp:103-112).  The elements are the FINE mesh's:
 *   domain_restr  [ne_fine][P]: the dofs (and orientations) of the PARENT of every fine element in the coarse space,
 *   range_restr   [ne_fine][P]: the fine space's own restriction,
 *   M [nmat][P][P], mat_id [ne_fine]: the local interpolation matrix (fine dof functionals of the child applied to the
 *                 parent's basis, mfem::FiniteElement::GetLocalInterpolation) of the child's place in its parent -- what
 *                 mfem::Mesh::GetRefinementTransforms() calls embeddings[e].matrix; nmat <= 256.
 * Every copy of a shared fine dof receives the same value (conforming spaces): one owner copy is stored, and the transpose
 * (the restriction of the V-cycle) reads through the same owner mask.  Mult / MultTranspose: pa_interp_mult[_transpose]. */
int pa_interp_create_refinement(pa_context *ctx, const pa_restriction_desc *domain_restr, const pa_restriction_desc *range_restr,
                                int nmat, const double *M, const uint8_t *mat_id, pa_halo *domain_halo, int n_true_domain,
                                int n_true_range, pa_interp **P);
int pa_interp_mult(pa_interp *P, const double *x_coarse, double *y_fine);
int pa_interp_mult_transpose(pa_interp *P, const double *x_fine, double *y_coarse);
void pa_interp_destroy(pa_interp *P);

#ifdef __cplusplus
}
#endif
#endif
