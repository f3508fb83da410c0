#include "comm.hpp"

#include <algorithm>
#include <cstdlib>

#include <dlfcn.h>

#include <cstring>
#include <string>

#include "pa_internal.hpp"

namespace palace {

namespace {

// RCCL is resolved lazily so that the single-GPU path has no load-time dependency on it.
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef int (*fn_get_id)(ncclUniqueId_t *);
typedef int (*fn_init_rank)(void **, int, ncclUniqueId_t, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_void)(void);
typedef const char *(*fn_errstr)(int);

struct Rccl {
  void *h = nullptr;
  fn_get_id GetUniqueId = nullptr;
  fn_init_rank CommInitRank = nullptr;
  fn_destroy CommDestroy = nullptr;
  fn_allreduce AllReduce = nullptr;
  fn_send Send = nullptr;
  fn_recv Recv = nullptr;
  fn_void GroupStart = nullptr, GroupEnd = nullptr;
  fn_errstr GetErrorString = nullptr;
  Rccl() {
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) throw pa::Error(std::string("cannot load librccl.so: ") + dlerror());
    auto sym = [&](const char *n) {
      void *p = dlsym(h, n);
      if (!p) throw pa::Error(std::string("missing RCCL symbol ") + n);
      return p;
    };
    GetUniqueId = (fn_get_id)sym("ncclGetUniqueId");
    CommInitRank = (fn_init_rank)sym("ncclCommInitRank");
    CommDestroy = (fn_destroy)sym("ncclCommDestroy");
    AllReduce = (fn_allreduce)sym("ncclAllReduce");
    Send = (fn_send)sym("ncclSend");
    Recv = (fn_recv)sym("ncclRecv");
    GroupStart = (fn_void)sym("ncclGroupStart");
    GroupEnd = (fn_void)sym("ncclGroupEnd");
    GetErrorString = (fn_errstr)sym("ncclGetErrorString");
  }
};
Rccl &rccl() {
  static Rccl r;
  return r;
}
constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values (rccl.h)

#define PA_NCCL(expr)                                                                           \
  do {                                                                                          \
    int rc__ = (expr);                                                                          \
    if (rc__ != 0) throw pa::Error(std::string(#expr) + " failed: " + rccl().GetErrorString(rc__)); \
  } while (0)

__global__ void k_pack(const double *__restrict__ v, const int32_t *__restrict__ idx, int n, double *__restrict__ buf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = v[idx[i]];
}
__global__ void k_unpack(double *__restrict__ v, const int32_t *__restrict__ idx, int n, const double *__restrict__ buf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[idx[i]] = buf[i];
}
// An owned dof may be shared with several neighbours: its index then appears once per neighbour
// and the adds must not race; contributions are applied neighbour by neighbour.
__global__ void k_unpack_add(double *__restrict__ v, const int32_t *__restrict__ idx, int n, const double *__restrict__ buf) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[idx[i]] += buf[i];
}
inline int blocks(int n) { return std::max(1, std::min(1024, (n + 255) / 256)); }

}  // namespace

void Comm::GetUniqueId(char *out) {
  ncclUniqueId_t id;
  PA_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(out, id.internal, kUniqueIdBytes);
}

Comm::Comm(int rank, int size, const char *unique_id) : rank_(rank), size_(size) {
  ncclUniqueId_t id;
  std::memcpy(id.internal, unique_id, kUniqueIdBytes);
  PA_NCCL(rccl().CommInitRank(&nccl_, size, id, rank));
}

Comm::Comm(int rank, LocalGroup &group) : rank_(rank), size_(group.Size()), local_(&group) {}

Comm::~Comm() {
  if (nccl_) rccl().CommDestroy(nccl_);
}

void LocalGroup::Arrive() {
  std::unique_lock<std::mutex> lk(m_);
  const long gen = generation_;
  if (++waiting_ == size_) {
    waiting_ = 0;
    generation_++;
    cv_.notify_all();
  } else {
    cv_.wait(lk, [&] { return generation_ != gen; });
  }
}

Halo::~Halo() {
  (void)hipFree(d_send_idx_), (void)hipFree(d_recv_idx_), (void)hipFree(d_sendbuf_), (void)hipFree(d_recvbuf_);
}

Halo::Halo(Comm &comm, int nnbr, const int *nbr, const int *send_off, const int32_t *send_idx, const int *recv_off,
           const int32_t *recv_idx)
    : comm_(&comm) {
  nbr_.assign(nbr, nbr + nnbr);
  send_off_.assign(send_off, send_off + nnbr + 1);
  recv_off_.assign(recv_off, recv_off + nnbr + 1);
  nsend_ = send_off_[nnbr], nrecv_ = recv_off_[nnbr];
  iface_.assign(send_idx, send_idx + nsend_);
  iface_.insert(iface_.end(), recv_idx, recv_idx + nrecv_);
  std::sort(iface_.begin(), iface_.end());
  iface_.erase(std::unique(iface_.begin(), iface_.end()), iface_.end());
  d_send_idx_ = pa::dev_upload(send_idx, (size_t)nsend_);
  d_recv_idx_ = pa::dev_upload(recv_idx, (size_t)nrecv_);
  {
    bool contiguous = nrecv_ > 0;
    for (int i = 1; i < nrecv_ && contiguous; i++) contiguous = recv_idx[i] == recv_idx[0] + i;
    const char *e = std::getenv("PALACE_AMD_HALO_INPLACE");
    if (contiguous && !(e && e[0] == '0')) recv_first_ = recv_idx[0];
  }
  const int nbuf = std::max(nsend_, nrecv_);
  d_sendbuf_ = pa::dev_alloc<double>((size_t)nbuf);
  d_recvbuf_ = pa::dev_alloc<double>((size_t)nbuf);
}

void Comm::AllReduceSum(double *d_buf, int n, hipStream_t s) {
  if (size_ == 1) return;
  if (local_) {  // values to the host, barrier, sum in rank order (the same on every rank), barrier, back to the device
    PA_REQUIRE(n <= LocalGroup::kMaxValues, "too many values for the in-process all-reduce");
    double *mine = local_->slots_.data() + (size_t)rank_ * LocalGroup::kMaxValues;
    PA_HIP(hipMemcpyAsync(mine, d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    local_->Arrive();
    std::vector<double> sum((size_t)n, 0.0);
    for (int r = 0; r < size_; r++)
      for (int i = 0; i < n; i++) sum[i] += local_->slots_[(size_t)r * LocalGroup::kMaxValues + i];
    local_->Arrive();
    PA_HIP(hipMemcpyAsync(d_buf, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, s));
    PA_HIP(hipStreamSynchronize(s));
    return;
  }
  PA_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat64, kNcclSum, nccl_, s));
}

void Comm::Barrier(hipStream_t s) {
  static double *d_one = pa::dev_alloc<double>(1);
  AllReduceSum(d_one, 1, s);
  PA_HIP(hipStreamSynchronize(s));
}

void Halo::ExchangeLocal(const double *sendbase, const std::vector<int> &send_off, double *recvbase,
                         const std::vector<int> &recv_off, hipStream_t s) const {
  LocalGroup &g = *comm_->local_;
  const int me = comm_->rank_;
  PA_HIP(hipStreamSynchronize(s));  // my send pieces are complete
  g.box_[me] = LocalGroup::Box{sendbase, nbr_.data(), send_off.data(), (int)nbr_.size()};
  g.Arrive();
  for (size_t k = 0; k < nbr_.size(); k++) {
    const int nr = recv_off[k + 1] - recv_off[k];
    if (!nr) continue;
    const LocalGroup::Box &b = g.box_[nbr_[k]];
    int j = 0;
    while (j < b.nnbr && b.nbr[j] != me) j++;
    PA_REQUIRE(j < b.nnbr && b.off[j + 1] - b.off[j] == nr, "halo plans of two ranks do not match");
    PA_HIP(hipMemcpyAsync(recvbase + recv_off[k], b.buf + b.off[j], sizeof(double) * nr, hipMemcpyDeviceToDevice, s));
  }
  PA_HIP(hipStreamSynchronize(s));
  g.Arrive();  // the send buffers may be reused
}

void Halo::Prolongate(double *d_lx, hipStream_t s) const {
  void *nccl_ = comm_->nccl_;
  if (nbr_.empty() && !comm_->local_) return;
  if (nsend_) hipLaunchKernelGGL(k_pack, dim3(blocks(nsend_)), dim3(256), 0, s, d_lx, d_send_idx_, nsend_, d_sendbuf_);
  if (comm_->local_) {
    ExchangeLocal(d_sendbuf_, send_off_, recv_first_ >= 0 ? d_lx + recv_first_ : d_recvbuf_, recv_off_, s);
    if (nrecv_ && recv_first_ < 0)
      hipLaunchKernelGGL(k_unpack, dim3(blocks(nrecv_)), dim3(256), 0, s, d_lx, d_recv_idx_, nrecv_, d_recvbuf_);
    PA_HIP(hipGetLastError());
    return;
  }
  PA_NCCL(rccl().GroupStart());
  for (size_t k = 0; k < nbr_.size(); k++) {
    const int ns = send_off_[k + 1] - send_off_[k], nr = recv_off_[k + 1] - recv_off_[k];
    if (ns) PA_NCCL(rccl().Send(d_sendbuf_ + send_off_[k], (size_t)ns, kNcclFloat64, nbr_[k], nccl_, s));
    double *dst = recv_first_ >= 0 ? d_lx + recv_first_ : d_recvbuf_;  // contiguous ghosts: no unpack pass
    if (nr) PA_NCCL(rccl().Recv(dst + recv_off_[k], (size_t)nr, kNcclFloat64, nbr_[k], nccl_, s));
  }
  PA_NCCL(rccl().GroupEnd());
  if (nrecv_ && recv_first_ < 0)
    hipLaunchKernelGGL(k_unpack, dim3(blocks(nrecv_)), dim3(256), 0, s, d_lx, d_recv_idx_, nrecv_, d_recvbuf_);
  PA_HIP(hipGetLastError());
}

void Halo::RestrictAdd(double *d_ly, hipStream_t s) const {
  void *nccl_ = comm_->nccl_;
  if (nbr_.empty() && !comm_->local_) return;
  // roles reversed: ghosts are packed and sent to their owners
  if (nrecv_ && recv_first_ < 0)
    hipLaunchKernelGGL(k_pack, dim3(blocks(nrecv_)), dim3(256), 0, s, d_ly, d_recv_idx_, nrecv_, d_sendbuf_);
  const double *src = recv_first_ >= 0 ? d_ly + recv_first_ : d_sendbuf_;  // contiguous ghosts: sent from the vector itself
  if (comm_->local_) {
    ExchangeLocal(src, recv_off_, d_recvbuf_, send_off_, s);
    for (size_t k = 0; k < nbr_.size(); k++) {
      const int nr = send_off_[k + 1] - send_off_[k];
      if (nr)
        hipLaunchKernelGGL(k_unpack_add, dim3(blocks(nr)), dim3(256), 0, s, d_ly, d_send_idx_ + send_off_[k], nr,
                           d_recvbuf_ + send_off_[k]);
    }
    PA_HIP(hipGetLastError());
    return;
  }
  PA_NCCL(rccl().GroupStart());
  for (size_t k = 0; k < nbr_.size(); k++) {
    const int ns = recv_off_[k + 1] - recv_off_[k], nr = send_off_[k + 1] - send_off_[k];
    if (ns) PA_NCCL(rccl().Send(src + recv_off_[k], (size_t)ns, kNcclFloat64, nbr_[k], nccl_, s));
    if (nr) PA_NCCL(rccl().Recv(d_recvbuf_ + send_off_[k], (size_t)nr, kNcclFloat64, nbr_[k], nccl_, s));
  }
  PA_NCCL(rccl().GroupEnd());
  for (size_t k = 0; k < nbr_.size(); k++) {
    const int nr = send_off_[k + 1] - send_off_[k];
    if (nr)
      hipLaunchKernelGGL(k_unpack_add, dim3(blocks(nr)), dim3(256), 0, s, d_ly, d_send_idx_ + send_off_[k], nr,
                         d_recvbuf_ + send_off_[k]);
  }
  PA_HIP(hipGetLastError());
}

}  // namespace palace
