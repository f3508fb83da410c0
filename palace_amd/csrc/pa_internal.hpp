// Internal declarations shared by the HIP translation units of libpalace_amd.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/palace_amd.h"
#include "pa_stream_host.hpp"  // kEssBit, kExclBit

struct pa_op;
typedef struct pa_op pa_op_fwd;

// CSR of a fully assembled local operator (pa_op_full_assemble), device arrays
struct pa_csr {
  bool symmetric = true;  // assembled from a symmetric operator (CsrOperator::MultTranspose relies on it)
  int32_t nrows = 0;
  int32_t ncols = 0;  // 0: square
  int64_t nnz = 0;
  int32_t *d_rowptr = nullptr, *d_col = nullptr;
  double *d_val = nullptr;
};

namespace pa {

void set_error(const std::string &msg);

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define PA_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t err__ = (expr);                                                               \
    if (err__ != hipSuccess)                                                                 \
      throw pa::Error(std::string(#expr) + " failed: " + hipGetErrorString(err__) + " (" +   \
                      __FILE__ + ":" + std::to_string(__LINE__) + ")");                      \
  } while (0)

#define PA_REQUIRE(cond, msg)                                                   \
  do {                                                                          \
    if (!(cond)) throw pa::Error(std::string(msg) + " [" #cond "]");            \
  } while (0)

// Every C entry point is wrapped: no exception crosses the ABI.
template <typename F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    set_error(e.what());
    return 1;
  } catch (...) {
    set_error("unknown error");
    return 1;
  }
}

template <typename T>
T *dev_alloc(size_t n) {
  T *p = nullptr;
  if (n == 0) n = 1;
  PA_HIP(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
  return p;
}

template <typename T>
T *dev_upload(const T *host, size_t n, hipStream_t s = nullptr) {
  T *p = dev_alloc<T>(n);
  if (n) PA_HIP(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
  PA_HIP(hipStreamSynchronize(s));
  return p;
}

constexpr int kExclBit16 = 1 << 15;  // flag in the slot permutation: this entry is the only copy of its dof
constexpr int kMaxP1 = 6;  // closed nodes p+1 <= 7
constexpr int kMaxQ1 = 7;

struct QData;

// Geometry factor data (fem/mesh.hpp:27-69): double[ne][11][Q] in HBM.
// Dense (non-tensor) element blocks keep it element-blocked for the MFMA kernel instead:
// double[ceil(ne/16)][11][Qpad][16] (eb == 16).
struct Geom {
  int ne = 0, q1d = 0, Q = 0;
  int eb = 0, Qpad = 0;
  int dim = 3, sdim = 3, nrows = 11;  // 2-D blocks: 6 rows {attr, w detJ, adj(J)^T/detJ (2x2)}; boundary (2 in 3): 8 rows
  double *d_qw = nullptr;         // quadrature weights (the q_w input of the 2-D curl-curl QFunctions)
  double *d_geom = nullptr;
  QData *metric = nullptr;        // lazily built G = J^T J [ne][6][Q] (tensor hex blocks), see QData
  int32_t *d_attr_e = nullptr;    // [ne] element attributes (metric form: coefficient lookup in the kernel)
  std::vector<int32_t> h_attr;    // host copy (tensor hex blocks), internal element order
  std::vector<int32_t> eorder;    // internal element p is the caller's element eorder[p] (empty: same order)
  std::vector<double> w1;         // 1-D quadrature weights (tensor hex blocks)
  // geometry from the nodes (hex27 blocks at four points per direction: the streaming curl-curl kernel's GEOMN form, round 5):
  // the 27 x 3 node coordinates of every element in internal element order, [ne padded to 4][27][3], and the 1-D geometry basis
  // at the quadrature points {B [4][3], G [4][3], weights [4]} (device)
  double *d_xnodes = nullptr, *d_gtab = nullptr;
  std::vector<double> wq;         // quadrature weights (dense blocks), host copy
  int refcount = 1;
};

// Parsed coefficient context (coeff_qf.h layout) living in device memory.
struct CoeffDev {
  const int32_t *attr_mat = nullptr;  // [nattr] (device) or nullptr when nattr == 0
  const double *mat = nullptr;        // [nmat * dim*dim] (device), column-major
  int nattr = 0;
};

struct CoeffHost {
  std::vector<int32_t> attr_mat;
  std::vector<double> mat;
  int dim = 3;
  size_t slots = 0;  // number of 8-byte slots this context occupied in the blob
  int32_t *d_attr_mat = nullptr;
  double *d_mat = nullptr;
  double *d_mat_t = nullptr;  // every matrix transposed; only when some matrix is not symmetric
  bool symmetric() const { return d_mat_t == nullptr; }
  // the device view the kernels read; inside a TransposeScope (pa_op_mult_transpose: A^T = B^T D(C^T) B for the
  // same trial / test evaluation) the transposed matrices
  CoeffDev dev() const;
};

// RAII switch for the launches of one transposed apply (host side, per thread)
struct TransposeScope {
  TransposeScope(bool on);
  ~TransposeScope();
  static bool active();

private:
  bool prev_;
};

// Packed symmetric pointwise operators (pre-assembled D): double[ne][ncomp][Q], the six upper
// entries of  w detJ adj^T C adj  (mass part, if any) followed by the six of  w detJ Jl^T C Jl
// (curl-curl part, if any).  Shared between an operator and its p-coarsened copies.
struct QData {
  double *d = nullptr;
  int ncomp = 0;
  int refcount = 1;
  // metric form (isotropic coefficients): H = (w / |detJ|) J^T J (six entries) and |detJ| / w per point, a
  // property of the mesh alone, shared by every operator and p-level on it.  Both pointwise operators follow
  // from it in registers:  (w / detJ) J^T c J = c H,   w detJ adj^T c adj = c (|detJ| / w) adj(H).
  bool metric = false;
  // Affine elements (round 6; four points per direction, the streaming H(curl) kernel): an element whose Jacobian is constant has
  // D(q) = w_q D_e -- 6 | 7 | 12 numbers per element instead of per point.  d_aff [ne padded to 4][2 ncomp] pairs of doubles: row
  // 2 c + h of element e = r_c {wz(2h), wz(2h + 1)} with r_c the point-independent factor of component c and wz the 1-D weight
  // along the lane's column (metric component 6, |detJ| / w: g {1 / wz(2h), 1 / wz(2h + 1)}); the kernel multiplies the result of
  // the D stage by the in-plane weight w(ta) w(tb).  batch_aff[b] != 0: the four elements of batch b are all affine (the
  // streaming kernel then reads 4 x 2 ncomp x 16 B instead of 4 x ncomp x 512 B).  The per-point data stays in place: every
  // other consumer (one-shot kernels, diagonal, assembly, the complex forms) is unaffected.  Built once per QData by
  // stream_affine_setup (pa_nd_hex_stream.hip) -- coarsened operators share it with the q-data.
  double *d_aff = nullptr;
  std::vector<unsigned char> batch_aff;
  bool aff_done = false;
  int n_aff_elems = 0, n_aff_batch_elems = 0;  // affine elements found / of them in all-affine batches (compressed)
  double wq2[2] = {0.0, 0.0};                  // the two distinct 1-D weights (symmetric four-point rule)
};

// Offset of component c at point q inside one element's block of packed q-data (H(curl) hexahedra).  In general
// [ncomp][Q]; with 4 points per direction the two points (qz, qz + 1) of a lane's column sit side by side,
// [ncomp][2][16][2], so the element kernels fetch 16 bytes per lane and instruction; with 5 points per direction
// [ncomp][126]: the pairs (qz 0, 1) and (qz 2, 3) of the 25 columns side by side, [2][25][2], then the points qz = 4, [25],
// and one pad entry that keeps every component 16-byte aligned.
__host__ __device__ inline int nd_qd_cstride(int q1d) { return q1d == 5 ? 126 : q1d * q1d * q1d; }
__host__ __device__ inline int nd_qd_offset(int q1d, int c, int q) {
  if (q1d == 4) return ((c * 2 + (q >> 5)) * 16 + (q & 15)) * 2 + ((q >> 4) & 1);
  if (q1d == 5) {
    const int qz = q / 25, t = q - 25 * qz;
    return c * 126 + (qz < 4 ? (qz >> 1) * 50 + 2 * t + (qz & 1) : 100 + t);
  }
  return c * q1d * q1d * q1d + q;
}

struct SubOp {
  Geom *geom = nullptr;
  QData *qd = nullptr;  // nullptr: matrix-free D from the geometry data (the reference default)
  int fe_type = 0, p = 0, q1d = 0, P = 0, Q = 0, ne = 0, lsize = 0;
  int qf = 0;
  uint32_t trial_ops = 0, test_ops = 0;
  int32_t *d_lidx = nullptr;  // [ne][P] signed tensor-order index: >=0 dof, <0 => -(1+dof) flipped
  // sorted order of E / E^T: entry m of element e is its m-th smallest global dof
  int32_t *d_sidx = nullptr;     // [ne][P] signed sorted index
  int32_t *d_sidx_bc = nullptr;  // copy with kEssBit on essential dofs (pa_op_set_essential)
  uint16_t *d_perm = nullptr;    // [ne][P] tensor-order slot of sorted entry m
  // Exclusive dofs (one element copy only, e.g. element interiors): stored straight into y by the
  // element kernel instead of going through the E-vector; the gather kernel then only visits the rest.
  uint16_t *d_perm_x = nullptr;  // perm with kExclBit16 on exclusive entries (set at finalize)
  int32_t *d_shared = nullptr;   // list of the dofs the gather kernel still has to sum
  int32_t *d_shared_bc = nullptr;  // the same with kEssBit on essential dofs (pa_op_set_essential)
  std::vector<int32_t> h_shared;
  std::vector<char> h_price_skip;  // pricing experiment only (PALACE_AMD_PRICE_BLOCK): E-vector entries taken off the gather
  int n_shared = 0;
  std::vector<uint16_t> h_perm;
  std::vector<int32_t> h_sidx;   // host copy (needed to build d_sidx_bc)
  // E^T as a gather (default): E-vector scratch and the CSR transpose of lidx
  double *d_ye = nullptr;      // [ne][P]
  double *d_ye2 = nullptr;     // second E-vector (two right-hand sides), allocated on first use
  int32_t *d_tptr = nullptr;   // [lsize + 1]
  int32_t *d_tent = nullptr;   // [ne * P] signed positions into d_ye
  // streaming form (pa_nd_hex_stream.hip): index words with the exclusive flag, byte slots, E^T of the shared dofs by runs
  uint32_t *d_idxc = nullptr;  // [ne][kIdxWords] run-compressed sorted element -> dof index (pa_stream_host.hpp)
  bool stream_default = true;  // false: the tables exist for the split-vector apply only, y = A x keeps the one-shot kernel (H1, p < 3)
  int32_t *d_blist[2] = {nullptr, nullptr};  // batch lists of the interior / interface phases (stream_set_interface)
  int n_blist[2] = {0, 0};
  bool has_blist = false;
  uint32_t *d_perm_s_bc = nullptr;                      // flag words with the essential dofs taken off the direct path
  std::vector<uint32_t> h_perm_s;
  int32_t *d_rhdr_bc = nullptr, *d_rpos_bc = nullptr;   // run list that also owns the essential rows
  int n_shared_bc = 0;
  uint32_t *d_perm_s = nullptr;                         // [ne][ceil(P/64)][16], four 8-bit tensor-order slots per word
  // four-point H(curl) kernel: the flag words on their own ([ne][16]; _bc: the copy with the essential dofs) and the slot words
  // through a dictionary -- elements whose sorted -> tensor-order permutation agrees share one entry (a structured mesh has a
  // handful: 10 on the 125 440-element bench cylinder), the element's entry number rides in a spare word of its index block
  uint32_t *d_flagw = nullptr, *d_flagw_bc = nullptr, *d_slots = nullptr;
  int n_slot_patterns = 0;
  double *d_coef_s = nullptr;                           // metric form: [ne][2] scalar mass / curl-curl coefficient per element
  int32_t *d_rhdr = nullptr, *d_rpos = nullptr;         // run headers {first dof | length - 1 | essential, first copy entry}; copy positions in d_ye
  int n_runs = 0, n_runs_bc = 0;
  uint32_t *d_rchunk = nullptr, *d_rchunk_bc = nullptr;  // [ceil(n_shared / 64)] RunChunk: run-start mask of 64 shared dofs + run / offset of the first
  // third form (round 6, stream_build_all): EVERY dof goes through the E-vector and is owned by the run gather -- no entry is
  // exclusive in this copy of the flag words -- so that the gather's epilogue can consume the result instead of storing it (the
  // fused Chebyshev step, launch_et_run_gather_step); essential dofs flagged as in the _bc copy
  uint32_t *d_flagw_all = nullptr, *d_rchunk_all = nullptr;
  uint32_t *d_perm_s_all = nullptr;  // the same for the five-point kernel, whose flags ride in the slot half-words
  int32_t *d_rhdr_all = nullptr, *d_rpos_all = nullptr;
  int n_all = 0, n_runs_all = 0;
  std::vector<char> h_ess_flag;  // the essential flags last fused (stream_set_essential), for stream_build_all
  std::vector<double> Bc, Gc, Bo;  // full 1-D tables [q1d][n] (host)
  double *d_tab = nullptr;        // the same on the device: [Bo | Bc | Gc]
  bool iso = false;               // every material coefficient is a multiple of the identity
  std::vector<uint8_t> ctx_blob;
  CoeffHost c0, c1;
};

// One sub-operator on a non-tensor element block (pa_dense.hip): dense tables on the FP64 matrix cores.
struct DenseSub {
  Geom *geom = nullptr;
  int fe_type = 0, P = 0, Q = 0, Qpad = 0, nch = 0, ne = 0, nb = 0, lsize = 0, KP = 0, PT = 0;
  int qf = 0, mode = 0;
  bool contra = false;  // plane H(div) mass: contravariant map in the vector-mass D (f_apply_hdiv_22)
  bool ye_rows = false;  // E-vector rows by element ([block][element][dof]) instead of by dof (pa_dense.hip: DenseArgs::ye_rows)
  uint32_t trial_ops = 0, test_ops = 0;
  int32_t *d_idx = nullptr;     // [nb][4 KP][16] signed index (oriented: <0 => -(1+dof) flipped); pads read zero
  int32_t *d_idx_bc = nullptr;  // copy with kEssBit on essential dofs
  uint8_t *d_ess_flag = nullptr;  // [lsize] 1 on essential dofs (row fix-up of the split-vector gather)
  uint16_t *d_co = nullptr;     // [nb][4 KP][16] packed int8 rows / columns of T_e (curl-oriented) or nullptr
  uint32_t *d_co2 = nullptr;    // the same, two dof slots per word: [nb][KP / 2][64] (resident kernel)
  int32_t *d_rows = nullptr;    // blocks that touch few dofs: the rows they have (sorted), n_rows of them; else nullptr
  int n_rows = 0;
  std::vector<int32_t> h_idx;
  std::vector<int32_t> h_off;  // plain [ne][P] offsets (full assembly)
  double *d_Tf = nullptr, *d_Tt = nullptr;  // MFMA A-operand fragments of the tables (forward / transposed)
  double *d_L = nullptr;       // LDS-resident form of the tables (fast path) or nullptr
  double *d_qdata = nullptr;   // packed pre-assembled D [nb][ncq][Qpad][16] (fast path)
  std::vector<uint16_t> h_co;  // host copy of d_co (restriction compare of the complex form)
  double chk_interp = 0.0, chk_deriv = 0.0;  // checksums of the basis tables (same)
  double *d_ye2 = nullptr;     // second E-vector (imaginary part of the complex form), allocated on first use
  uint8_t *d_affine = nullptr; // [nb] blocks whose elements all have a constant Jacobian (D_q = (w_q / w_0) D_0) or nullptr
  double *d_wrel = nullptr;    // [Q4] w_q / w_0
  int n_affine = 0;
  int32_t *d_blist[2] = {nullptr, nullptr};  // mixed meshes: the affine blocks / the others (else nullptr: one kind only)
  int n_blist[2] = {0, 0};
  int L_rows = 0, ncq = 0, num_cu = 0;
  double *d_interp = nullptr, *d_deriv = nullptr;  // plain tables (diagonal assembly)
  int32_t *d_off = nullptr;    // plain [ne][P] offsets (diagonal assembly)
  int8_t *d_cor = nullptr;     // plain curl_orients or nullptr
  uint8_t *d_ori = nullptr;
  double *d_ye = nullptr;      // E-vector [nb][4 KP][16]
  int32_t *d_tptr = nullptr, *d_tent = nullptr;
  // run form of the same map (pa_stream_host.hpp: build_runs_dense): chunk masks, run headers, one position per run and copy
  uint32_t *d_rchunk = nullptr, *d_rpos_run = nullptr;
  int32_t *d_rhdr = nullptr;
  std::vector<uint8_t> ctx_blob;
  CoeffHost c0, c1;
};

// pa_mixed.hip: one space of a mixed-space operator (plain [ne][P] layouts) and the operator / error integrator itself
struct MixedSide {
  int fe_type = 0, P = 0, lsize = 0, nc = 3;
  std::vector<int32_t> h_off;  // [ne][P] L-vector indices (full assembly)
  int32_t *d_sidx = nullptr;
  int8_t *d_cor = nullptr;
  double *d_tabF = nullptr, *d_tabT = nullptr;
  int32_t *d_tptr = nullptr, *d_tent = nullptr;
};
struct MixedSub {
  Geom *geom = nullptr;
  int ne = 0, Q = 0, qf = 0, kind = 0;
  bool error = false;
  MixedSide s1, s2;  // apply: trial, test; error: first and second input
  CoeffHost c0, c1;
  double *d_ye = nullptr;
  mutable double *d_ye_t = nullptr;  // E-vector of the transposed apply (trial side), allocated at its first use
};

void parse_coeff(const void *blob, size_t bytes, int dim, CoeffHost &out, size_t slot_offset);

// kernels (pa_geom.hip, pa_nd_hex.hip, pa_h1_hex.hip)
void launch_geom(const pa_mesh_desc &mesh, Geom &g, hipStream_t s);
void launch_nd_hex_apply(const SubOp &so, const double *x, double *y, double *ye, bool masked, hipStream_t s,
                         bool accumulate = true, int ess_policy = -1, const double *x1 = nullptr, double *y1 = nullptr,
                         double *ye1 = nullptr);
bool nd_hex_supports_two_rhs(const SubOp &so);
void launch_et_gather2(const SubOp &so, double *y0, double *y1, bool accumulate, hipStream_t s, const double *x0,
                       const double *x1, int ess_policy);
bool nd_hex_fuses_essential(const SubOp &so);
void finalize_exclusive(pa_op_fwd *op);
void launch_et_gather(const SubOp &so, double *y, bool accumulate, hipStream_t s, const double *x = nullptr,
                      int ess_policy = -1);
void launch_nd_hex_qdata(SubOp &so, hipStream_t s);
void launch_nd_hex_metric(SubOp &so, hipStream_t s);
void launch_nd_hex_diag(const SubOp &so, double *diag, hipStream_t s);
// pa_nd_hex_stream.hip
struct SplitIO;
bool h1_hex_stream_ok(const SubOp &so);       // the streaming form is the default for y = A x
bool h1_hex_stream_capable(const SubOp &so);  // ... its tables can be built (split-vector applies use it at every order)
void launch_h1_hex_stream(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, const SplitIO *split = nullptr,
                          bool all = false);
bool nd_hex_stream_ok(const SubOp &so);
void build_stream(SubOp &so);
void stream_set_essential(SubOp &so, const std::vector<char> &flag);
void free_stream(SubOp &so);
// Split vectors of a multi-rank apply without L-vector copies: local dofs [0, n_true) are read from x and written to y,
// the ghosts [n_true, lsize) are read from xg0 or xg1 (the parity of the device-resident counter *sel picks the buffer; sel ==
// NULL: xg0) and written to yg (all three unshifted: entry 0 is local dof n_true)
struct SplitIO {
  int n_true;
  const double *xg0, *xg1;
  const unsigned long long *sel;
  double *yg;
};
bool nd_hex_stream_split_ok(const SubOp &so);
void launch_nd_hex_stream(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase = -1,
                          const SplitIO *split = nullptr);
void stream_set_interface(SubOp &so, const std::vector<char> &flag);
// One step of a smoother recurrence consumed inside E^T (round 6): with t = (A x)[d] (the essential rows fixed as in the masked
// apply) the run gather writes  out[d] (+)= x[d] + sd (x[d] - ep[d]) + sr dinv[d] (r0[d] - t)  for every dof and never stores t:
// the Chebyshev step of chebyshev.cpp:204-218 in its accumulated form (linalg.hip: OpChebStep3) without the round trip of A x.
// mode 2, the residual form: with r = r0[d] - t the gather writes  res[d] = r  (if res) and  out[d] = sr dinv[d] r  (if out) --
// r = b - A y of gmg.cpp:186-188 / chebyshev.cpp:196-200 and the first direction d_0 = c_0 D^-1 r of the polynomial.
struct GatherStep {
  double sd, sr;
  const double *dinv, *r0, *ep;  // ep == nullptr: e_{k-1} = 0
  double *out;
  int add;
  double *res;
  int mode;  // 1: Chebyshev step, 2: residual
  // multi-rank (split-vector) form: owned dofs other ranks hold as ghosts (bit 2 of iface_mask[d]) are not finished by this rank's
  // gather -- their sum goes to t_iface[d] and the halo kernel that adds the neighbours' rows applies the step (comm.hip:
  // RestrictAddDirectStep); ghost rows go to the ghost output as in the plain split form.  nullptr: one rank.
  const unsigned char *iface_mask;
  double *t_iface;
  // operators with further (surface) blocks beside the one whose gather runs the step: their contributions, accumulated beforehand,
  // are added to the sum of every row that is not an essential one (nullptr: none).  Dense path only.
  const double *t_add;
};
void launch_et_run_gather2(const SubOp &so, double *y, double *y1, hipStream_t s, const double *x, const double *x1, bool masked,
                           int ess_policy, const double *ye1);
bool stream_build_all(SubOp &so);  // the index copies the fused step needs (idempotent; false: this block has no such form)
void launch_nd_hex_stream_all(const SubOp &so, const double *x, hipStream_t s, const SplitIO *split = nullptr);
void launch_et_run_gather_step(const SubOp &so, const double *x, const GatherStep &step, int ess_policy, hipStream_t s,
                               const SplitIO *split = nullptr);
bool nd_hex_stream5_ok(const SubOp &so);
void launch_nd_hex_stream5(const SubOp &so, const double *x, double *y, bool masked, hipStream_t s, int phase,
                           const SplitIO *split = nullptr, bool all = false);
void launch_nd_hex_stream5_complex(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                                   double *ye_i, bool masked, hipStream_t s);
void launch_et_run_gather(const SubOp &so, double *y, bool accumulate, hipStream_t s, const double *x, bool masked,
                          int ess_policy, const double *ye = nullptr, const SplitIO *split = nullptr);
void stream_element_coefficients(SubOp &so);
bool nd_hex_stream_complex_ok(const SubOp &sr, const SubOp &si);
void launch_nd_hex_stream_complex(const SubOp &sr, const SubOp &si, const double *xr, const double *xi, double *yr, double *yi,
                                  double *ye_i, bool masked, hipStream_t s);
void launch_h1_hex_apply(const SubOp &so, const double *x, bool masked, hipStream_t s);
void launch_h1_hex_qdata(SubOp &so, hipStream_t s);
void launch_h1_hex_diag(const SubOp &so, double *diag, hipStream_t s);

// pa_dense.hip
void launch_geom_dense(const pa_mesh_dense_desc &mesh, Geom &g, hipStream_t s);
DenseSub *make_dense_sub(pa_geom *geom, const pa_restriction_desc &r, const pa_dense_basis_desc &b, int qf,
                         const void *ctx, size_t ctx_size, uint32_t trial_ops, uint32_t test_ops, int height, bool contra = false);
void free_dense_sub(DenseSub *ds);
void dense_set_essential(DenseSub &ds, const std::vector<char> &flag);
void launch_dense_apply(const DenseSub &ds, const double *x, bool masked, hipStream_t s, const SplitIO *split = nullptr);
// split: rows >= n_true go to split->yg; ess_policy >= 0 (with split only): essential rows are set to x (1) or 0 (0) here
void launch_dense_gather(const DenseSub &ds, double *y, bool accumulate, hipStream_t s, const double *ye = nullptr,
                         const SplitIO *split = nullptr, const double *x = nullptr, int ess_policy = -1);
bool dense_split_ok(const DenseSub &ds);
bool dense_complex_ok(const DenseSub &dr, const DenseSub &di);
void launch_dense_gather_signed(const DenseSub &ds, double *y, double sign, bool skip_ess, hipStream_t s);
// the fused smoother step / residual on a dense block (round 6): the CSR-form gather owns every row; its epilogue consumes the sum
bool dense_fused_step_ok(const DenseSub &ds);
double time_dense_gather(const DenseSub &ds, const int32_t *d_tent);  // ms per launch of the plain E^T gather with this map (set-up)
int dense_gather_group(const DenseSub &ds);
void launch_zero_rows(double *v, const int32_t *rows, int n, hipStream_t s);  // v[rows[i]] = 0
constexpr int kMaxSurfaceBlocks = 8;
struct SurfaceYe {
  const double *ye[kMaxSurfaceBlocks];
};
// out[rows[i]] = sum over the entries of row i (block blk[e], signed position ent[e]) -- the surface blocks of a dense operator in one launch
void launch_surface_rows(const int32_t *rows, int n, const int32_t *row_ptr, const int32_t *ent, const uint8_t *blk, const SurfaceYe &ye,
                         double *out, hipStream_t s);  // lanes per dof of the CSR-form gather (pa_dense.hip: et_gather_group_kernel)
void launch_dense_gather_step(const DenseSub &ds, const double *x, const GatherStep &step, int ess_policy, hipStream_t s,
                              const SplitIO *split = nullptr);
void launch_dense_complex(const DenseSub &dr, const DenseSub &di, const double *xr, const double *xi, double *ye_i, hipStream_t s,
                          bool masked = false);
void launch_dense_diag(const DenseSub &ds, double *diag, hipStream_t s);
void launch_et_gather_raw(int n, const int32_t *tptr, const int32_t *tent, const double *ye, double *y,
                          bool accumulate, hipStream_t s, const int32_t *list = nullptr, const double *x = nullptr,
                          int ess_policy = -1);

// pa_mixed.hip
MixedSub *make_mixed_sub(pa_geom *geom, const pa_restriction_desc &r1, const pa_dense_basis_desc &b1,
                         const pa_restriction_desc &r2, const pa_dense_basis_desc &b2, int qf, const void *ctx,
                         size_t ctx_size);
MixedSub *make_mixed_gradient_sub(pa_geom *geom, const pa_restriction_desc &r1, const pa_dense_basis_desc &b1,
                                  const pa_restriction_desc &r2, const pa_dense_basis_desc &b2, int comp_stride, int qf,
                                  const void *ctx, size_t ctx_size);
void free_mixed_sub(MixedSub *ms);
void launch_mixed_apply(const MixedSub &ms, const double *x, double *y, bool accumulate, hipStream_t s, bool transpose = false);
void launch_mixed_error(const MixedSub &ms, const double *u1, const double *u2, double *out, hipStream_t s);

}  // namespace pa

struct pa_geom : pa::Geom {};

struct pa_op {
  uint64_t id = 0;  // unique per operator object (pa_op_create / coarsen)
  mutable uint64_t cplx_partner = 0;  // id of the operator the complex-form check below was last made against, and its answer
  mutable int cplx_ok = 0;
  int height = 0, width = 0;
  bool finalized = false;
  bool has_essential = false;
  std::vector<int32_t> ess_sorted;  // the list fused into the index tables (pa_op_set_essential)
  bool symmetric() const;  // every coefficient matrix of every sub-operator is symmetric => A^T = A
  std::vector<pa::SubOp *> subs;
  std::vector<pa::DenseSub *> dsubs;
  std::vector<pa::MixedSub *> msubs;  // trial space != test space (pa_op_add_sub_dense_mixed)
  // fused smoother step of a dense operator with surface blocks (round 6, pa_op_prepare_fused_step): the surface blocks accumulate into
  // t_extra (zero outside their rows; the union of their rows is reset before every use), the volume block's gather adds it
  double *d_t_extra = nullptr;
  int32_t *d_extra_rows = nullptr;
  int n_extra_rows = 0;
  // ... by ONE gather over the union of their rows (entries of every surface block, block by block: the order of the accumulating
  // gathers it replaces): row pointers, signed E-vector positions, block numbers
  int32_t *d_urow_ptr = nullptr, *d_uent = nullptr;
  uint8_t *d_ublk = nullptr;
};
