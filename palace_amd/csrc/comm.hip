#include "comm.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>

#include <dlfcn.h>

#include <cstring>
#include <limits>
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
// blocks that touch a mailbox: few, walking it with a stride (each of them bumps a counter before the flags go up)
// blocks of the exchange kernels: one round of their unrolled loops per thread where the two-level block counter (last_block)
// keeps the completion detection cheap -- 4 entries per thread in the send / unpack kernels, 2 in the summing one
inline int mail_blocks(int n) { return std::max(1, std::min(256, (n + 1023) / 1024)); }
inline int sum_blocks(int n) { return std::max(1, std::min(512, (n + 511) / 512)); }

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
  if (d_one_) (void)hipFree(d_one_);
  for (size_t r = 0; r < remote_.size(); r++)
    if (remote_ipc_[r]) (void)hipIpcCloseMemHandle(remote_[r]);
  if (d_remote_) (void)hipFree(d_remote_);
  if (setup_stream_) (void)hipStreamDestroy(setup_stream_);
  if (arena_) (void)hipFree(arena_);
  if (h_err_) (void)hipHostFree(h_err_);
}

std::vector<double> Comm::SetupGather(double mine, hipStream_t s) {
  PA_REQUIRE(PeerReady() && size_ <= kMaxReduceSetup, "set-up gather: peer transport only");
  std::vector<double> v((size_t)size_, 0.0);
  v[(size_t)rank_] = mine;
  if (size_ == 1) return v;
  double *d = pa::dev_upload(v.data(), v.size(), s);
  PeerAllReduce(d, size_, s, 1);  // (every rank contributes zeros outside its own entry: the sum is the gather)
  PA_HIP(hipMemcpyAsync(v.data(), d, sizeof(double) * v.size(), hipMemcpyDeviceToHost, s));
  PA_HIP(hipStreamSynchronize(s));
  (void)hipFree(d);
  PeerCheckNow();
  return v;
}

int Comm::RanksOnMyDevice(hipStream_t s) {
  if (ranks_on_device_ > 0) return ranks_on_device_;
  if (size_ == 1 || local_ || !PeerReady() || size_ > kMaxReduceSetup) {
    // one rank; the threads of an in-process group (one device by construction); no set-up channel: assume the worst
    ranks_on_device_ = (size_ == 1) ? 1 : size_;
    return ranks_on_device_;
  }
  int dev = 0;
  hipDeviceProp_t prop{};
  double id = -1.0;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    id = (double)(((long long)prop.pciDomainID << 24) | ((long long)(prop.pciBusID & 0xff) << 16) | ((long long)(prop.pciDeviceID & 0xff) << 8)) + 1.0;
  const std::vector<double> ids = SetupGather(id, s);
  int same = 0;
  for (const double v : ids) same += (id < 0.0 || v < 0.0 || v == id) ? 1 : 0;  // (an unknown identity counts as shared)
  ranks_on_device_ = std::max(1, same);
  return ranks_on_device_;
}

std::vector<double> Comm::AllGatherVHost(const std::vector<double> &mine, std::vector<long long> *offsets) {
  PA_REQUIRE(PeerReady() && size_ > 1, "all-gather-v over the peer transport: not connected");
  const std::vector<double> cnt = SetupGather((double)mine.size(), setup_stream_);
  std::vector<long long> off((size_t)size_ + 1, 0);
  long long longest = 0;
  for (int r = 0; r < size_; r++) {
    off[(size_t)r + 1] = off[(size_t)r] + (long long)cnt[(size_t)r];
    longest = std::max(longest, (long long)cnt[(size_t)r]);
  }
  PA_REQUIRE(off[(size_t)size_] < (1ll << 31), "all-gather-v: too many entries");
  std::vector<double> all((size_t)off[(size_t)size_], 0.0);
  std::copy(mine.begin(), mine.end(), all.begin() + off[(size_t)rank_]);
  if (offsets) *offsets = off;
  if (longest == 0) return all;
  const long long chunk = std::min<long long>(longest, 1ll << 19);  // doubles per rank and round (4 MB)
  const size_t bytes = sizeof(double) * (size_t)chunk;
  const size_t block = PeerAlloc(bytes);
  // every rank's block offset (blocks of destroyed plans are reused rank by rank: the offsets may differ)
  const std::vector<double> blocks = SetupGather((double)block, setup_stream_);
  try {
    for (long long c0 = 0; c0 < longest; c0 += chunk) {
      const long long n_mine = std::max(0ll, std::min(chunk, (long long)mine.size() - c0));
      if (n_mine) PA_HIP(hipMemcpyAsync(arena_ + block, mine.data() + c0, sizeof(double) * (size_t)n_mine, hipMemcpyHostToDevice, setup_stream_));
      PA_HIP(hipStreamSynchronize(setup_stream_));  // (the arena holds the chunk before the barrier releases the readers)
      (void)SetupGather(0.0, setup_stream_);
      for (int r = 0; r < size_; r++) {
        if (r == rank_) continue;
        const long long n_r = std::max(0ll, std::min(chunk, (long long)cnt[(size_t)r] - c0));
        if (n_r)
          PA_HIP(hipMemcpy(all.data() + off[(size_t)r] + c0, remote_[(size_t)r] + (size_t)blocks[(size_t)r], sizeof(double) * (size_t)n_r,
                           hipMemcpyDeviceToHost));
      }
      (void)SetupGather(0.0, setup_stream_);  // every rank has read this round: the blocks may be overwritten
    }
  } catch (...) {
    PeerFree(block, bytes);
    throw;
  }
  PeerFree(block, bytes);
  return all;
}

void LocalGroup::Arrive() {
  std::unique_lock<std::mutex> lk(m_);
  PA_REQUIRE(!aborted_, "in-process rank group aborted: another rank failed");
  const long gen = generation_;
  if (++waiting_ == size_) {
    waiting_ = 0;
    generation_++;
    cv_.notify_all();
  } else {
    // a rank that throws between two barriers calls Abort() (pa_local_group_abort) and releases the others; a rank that
    // simply never arrives is caught by the time limit
    const bool ok = cv_.wait_for(lk, std::chrono::seconds(timeout_s_), [&] { return generation_ != gen || aborted_; });
    if (!ok) aborted_ = true, cv_.notify_all();
    PA_REQUIRE(ok, "in-process rank group: a rank did not reach the barrier in time");
    PA_REQUIRE(generation_ != gen, "in-process rank group aborted: another rank failed");
  }
}
void LocalGroup::Abort() {
  std::lock_guard<std::mutex> lk(m_);
  aborted_ = true;
  cv_.notify_all();
}

Halo::~Halo() {
  (void)hipFree(d_send_idx_), (void)hipFree(d_recv_idx_), (void)hipFree(d_sendbuf_), (void)hipFree(d_recvbuf_);
  if (d_ghost_out_) (void)hipFree(d_ghost_out_);
  FreePeer();
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
  send_min_ = recv_min_ = std::numeric_limits<int>::max(), send_max_ = recv_max_ = -1;
  for (int i = 0; i < nsend_; i++) send_min_ = std::min(send_min_, send_idx[i]), send_max_ = std::max(send_max_, send_idx[i]);
  for (int i = 0; i < nrecv_; i++) recv_min_ = std::min(recv_min_, recv_idx[i]), recv_max_ = std::max(recv_max_, recv_idx[i]);
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
  // direct stores into the neighbours' mailboxes where the ranks' arenas are connected (PALACE_AMD_HALO=rccl: the RCCL
  // send / receive groups -- or, in an in-process group, the host-barrier copies -- instead)
  const char *mode = std::getenv("PALACE_AMD_HALO");
  // (one rank whose plan names itself as the neighbour -- the per-rank cost proxy of scripts/time_halo_mult.py -- included)
  if (comm.PeerReady() && (comm.Size() > 1 || nnbr > 0) && !(mode && std::string(mode) == "rccl")) PeerSetup(send_idx);
}

void Halo::Validate(int n_true, int n_local) const {
  PA_REQUIRE(send_min_ >= 0 && send_max_ < n_true, "halo plan: a dof to send is not a true dof of this vector");
  PA_REQUIRE(recv_min_ >= n_true && recv_max_ < n_local, "halo plan: a ghost slot lies outside [n_true, n_local)");
  if (recv_first_ >= 0)
    PA_REQUIRE(recv_first_ >= n_true && recv_first_ + nrecv_ <= n_local, "halo plan: in-place ghost range outside the vector");
}

void Comm::AllReduceSum(double *d_buf, int n, hipStream_t s) {
  if (size_ == 1) return;
  if (PeerReady() && (!nccl_ || !std::getenv("PALACE_AMD_PEER_NO_REDUCE"))) {
    // (more values than one message holds: in pieces -- a communicator without RCCL has no other way)
    for (int i0 = 0; i0 < n; i0 += kMaxReduce) PeerAllReduce(d_buf + i0, std::min(kMaxReduce, n - i0), s);
    return;
  }
  if (local_) {  // values to the host, barrier, sum in rank order (the same on every rank), barrier, back to the device
    if (n > LocalGroup::kMaxValues) {  // (a whole Gram-Schmidt column: in pieces, like the peer transport above)
      for (int i0 = 0; i0 < n; i0 += LocalGroup::kMaxValues) AllReduceSum(d_buf + i0, std::min(LocalGroup::kMaxValues, n - i0), s);
      return;
    }
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
  PA_REQUIRE(nccl_, "global sum: this communicator has neither RCCL nor a connected peer transport");
  PA_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat64, kNcclSum, nccl_, s));
}

void Comm::Barrier(hipStream_t s) {
  if (!d_one_) {
    d_one_ = pa::dev_alloc<double>(1);
    PA_HIP(hipMemset(d_one_, 0, sizeof(double)));
  }
  // peer transport: the set-up channel (own sequence counter, flags and slots), so that a barrier on another stream than the
  // solvers' (PeerSetup) never touches the counters of all-reduces still queued there
  if (size_ > 1 && PeerReady())
    PeerAllReduce(d_one_, 1, s, 1);
  else
    AllReduceSum(d_one_, 1, s);
  PA_HIP(hipStreamSynchronize(s));
  PeerCheckNow();
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
  if (peer_) return PeerExchange(0, d_lx, s);
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
  if (peer_) return PeerExchange(1, d_ly, s);
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

// ---- peer transport -------------------------------------------------------------------------------------------------------
namespace {

constexpr unsigned long long kDescMagic = 0x70616c6163654844ull;
// arena layout (bytes): error word | all-reduce sequence | all-reduce flags [kMaxRanks] | all-reduce slots
// [2][kMaxRanks][kMaxReduce] | halo descriptors [kMaxHalos] | mailboxes ...
// two all-reduce channels: 0 = the solvers' global sums (context stream), 1 = set-up barriers (Comm::Barrier)
constexpr size_t kOffArSeq[2] = {8, 16};
constexpr size_t kOffArFlags[2] = {64, 64 + 8 * Comm::kMaxRanks};
constexpr size_t kOffArSlots[2] = {2048, 2048 + 2ull * Comm::kMaxRanks * Comm::kMaxReduce * sizeof(double)};
constexpr int kArMaxN[2] = {Comm::kMaxReduce, Comm::kMaxReduceSetup};
constexpr size_t kArSlotBytes = 2ull * Comm::kMaxRanks * (Comm::kMaxReduce + Comm::kMaxReduceSetup) * sizeof(double);
static_assert(kOffArFlags[1] + 8 * Comm::kMaxRanks <= kOffArSlots[0], "arena header layout");
constexpr size_t kOffDesc = ((kOffArSlots[0] + kArSlotBytes + 4095) / 4096) * 4096;

struct HaloDesc {
  unsigned long long ready;  // kDescMagic + plan id once the owner has filled it (a slot is reused by plan id + kMaxHalos)
  int nnbr, nrecv, nsend, pad;
  int nbr[Comm::kMaxNbr], recv_off[Comm::kMaxNbr + 1], send_off[Comm::kMaxNbr + 1];
  unsigned long long off_mb[2], off_local;  // byte offsets in the owner's arena: mailboxes of P / P^T, its PeerLocal
};
constexpr size_t kOffDynamic = ((kOffDesc + sizeof(HaloDesc) * Comm::kMaxHalos + 4095) / 4096) * 4096;

// flags of one halo in its owner's arena (dir 0: P, owners -> ghosts; 1: P^T, ghosts -> owners)
struct PeerLocal {
  unsigned long long flag[2][Comm::kMaxNbr];  // written by neighbour k: the sequence number of the message it has delivered
  unsigned long long ack[2][Comm::kMaxNbr];   // written by neighbour k: the last message of mine it has consumed
};
// my own counters: ordinary (cached) device memory -- a thousand blocks bumping a counter in uncached memory cost more than
// the exchange itself (measured: 14 us per kernel)
struct PeerCounters {
  unsigned long long seq[2];      // my exchange counters (advanced on the device: graphs replay)
  // block counters of the send / consume kernels, two levels: a block bumps the counter of its group (block index mod 16, each
  // in its own 64-byte line), the last of a group bumps `top` -- a single counter costs 13 ns per block (measured), which capped
  // the kernels at 64 blocks and left their threads several dependent round trips to uncached memory each
  struct BlockCounter {
    unsigned int top, pad[15];
    unsigned int grp[16][16];
  } done[2], cons[2];
};
// one neighbour as the kernels see it
struct PeerNbr {
  double *dst[2];               // where my piece starts in the neighbour's mailbox of direction d (buffer 0)
  long long stride[2];          // doubles between the neighbour's two buffers
  unsigned long long *flag[2];  // my slot of the neighbour's PeerLocal::flag / ack
  unsigned long long *ack[2];
  int off[2], n[2];             // my pieces: what I send in P (send list) / in P^T (ghosts)
  int rn[2];                    // what I receive from it in P / P^T
};

// wall_clock64 runs at 100 MHz.  The limit only exists so that a lost message cannot hang the GPU for good; it has to be
// longer than any skew between ranks (host-side set-up of a coarse solver, graph instantiation, rank-0 I/O): one minute by
// default, PALACE_AMD_PEER_TIMEOUT_S.  A rank whose wait has timed out raises its error word (host memory: every host
// synchronisation point of the library looks at it) and gives up on every later wait at once instead of queueing minutes of
// further time-outs.
__device__ long long g_spin_ticks = 6000000000ll;
__device__ int g_peer_fence = 0;  // 1: system-scope release / acquire fences (Comm::SetFenced)

// Memory ordering without system-scope fences.  A release / acquire fence at system scope writes back / invalidates the
// whole L2 -- with megabytes of freshly written vector data in it that costs more than the exchange (measured: every block
// of the first version did one).  The arena is uncached (or fine-grained) memory and every access to it below is a relaxed
// system-scope atomic, i.e. a plain load / store that bypasses the caches; what remains to be ordered is "my data stores
// have been performed before the flag store is issued": s_waitcnt vmcnt(0) (stores are acknowledged once they are performed)
// and the block counter in front of the flag stores.
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys_f64(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_sys_f64(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void stores_performed() {
  if (g_peer_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // (system scope: L2 write-back, then the wait)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void messages_arrived() {
  if (g_peer_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// waits until *p >= want; gives up (and raises *err) after g_spin_ticks so that a lost message cannot hang the GPU
__device__ bool spin_ge(const unsigned long long *p, unsigned long long want, unsigned long long *err) {
  if (ld_sys(p) >= want) return true;
  const long long t0 = wall_clock64();
  for (unsigned it = 1;; it++) {
    if (ld_sys(p) >= want) return true;
    if ((it & 255u) == 0) {
      if (ld_sys(err)) return false;  // this rank has already given up on a wait
      if (wall_clock64() - t0 > g_spin_ticks) {
        st_sys(err, 1ull);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// true in the block that finishes last of `nblocks` (every one's stores have been performed by then)
__device__ __forceinline__ bool last_block(PeerCounters::BlockCounter *c, const unsigned int nblocks) {
  __shared__ bool last;
  stores_performed();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int g = blockIdx.x & 15u, ng = (nblocks - g + 15u) / 16u, ngroups = nblocks < 16u ? nblocks : 16u;
    bool l = false;
    if (atomicAdd(&c->grp[g][0], 1u) == ng - 1u) {
      c->grp[g][0] = 0;  // (every block of the group has arrived: free for the next launch)
      l = atomicAdd(&c->top, 1u) == ngroups - 1u;
      if (l) c->top = 0;
    }
    last = l;
  }
  __syncthreads();
  return last;
}

// every block waits (its first threads, one per neighbour) for the messages of exchange s of direction dir, and for the
// acknowledgement of my message s - 1 (so that message s + 1 may overwrite its buffer)
__device__ __forceinline__ void wait_exchange(const PeerNbr *nb, const int nnbr, const PeerLocal *L, const int dir,
                                              const unsigned long long s, unsigned long long *err) {
  for (int k = threadIdx.x; k < nnbr; k += blockDim.x) {
    if (nb[k].rn[dir] > 0) spin_ge(&L->flag[dir][k], s, err);
    if (nb[k].n[dir] > 0 && s > 0) spin_ge(&L->ack[dir][k], s - 1ull, err);
  }
  messages_arrived();
  __syncthreads();
}

// (1) my pieces of direction `dir` into the neighbours' mailboxes, then their flags.  Source v[idx[i]] (idx == nullptr:
// v[first + i]).  FUSED (P only, ParOperator::Mult): the vector is built on the way -- lx[i] = mask[i] & 1 ? 0 : x[i] for the
// true dofs, the same masked values into the mailboxes (rap.cpp:207-216: tx = x, tx[ess] = 0, lx = P tx).
// MODE 2 (direct form of ParOperator::Mult, P only): the masked values straight from x, no L-vector; the last block then waits
// for the neighbours' messages of this exchange, so that the kernel behind this one may read the mailbox in place.
// ack_p (P^T of the direct form): the last block also acknowledges the P messages of this Mult -- the element kernel that
// read them from the mailbox has finished by now.
template <int MODE>
__global__ __launch_bounds__(256) void k_peer_send(const PeerNbr *__restrict__ nb, const int nnbr, PeerCounters *__restrict__ L,
                                                    const int dir, const int total, const double *__restrict__ v,
                                                    const int32_t *__restrict__ idx, const int first,
                                                    const uint8_t *__restrict__ mask, double *__restrict__ lx, const int n_true,
                                                    const int mblk, const PeerLocal *__restrict__ F, unsigned long long *err,
                                                    const int ack_p) {
  constexpr bool FUSED = MODE == 1;
  constexpr bool MASKED = MODE != 0;
  // Blocks [0, mblk) walk the mailbox entries (and are the only ones counted before the flags go up: a block counter that
  // thousands of blocks bump costs 13 ns per block, measured -- more than the whole exchange); FUSED: the blocks behind
  // them copy the true dofs, one per thread.
  if (FUSED && (int)blockIdx.x >= mblk) {
    const int d = ((int)blockIdx.x - mblk) * blockDim.x + threadIdx.x;
    if (d < n_true) lx[d] = (mask[d] & 1) ? 0.0 : v[d];
    return;
  }
  const unsigned long long s = L->seq[dir] + 1ull;
  const int par = (int)(s & 1ull);
  // four entries per thread and round, their loads side by side (a stride loop of dependent loads is latency-bound)
  const int stride = mblk * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
    int d[4];
    double val[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * stride;
      d[j] = i < total ? (idx ? idx[i] : first + i) : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) val[j] = d[j] >= 0 ? ((MASKED && (mask[d[j]] & 1)) ? 0.0 : v[d[j]]) : 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * stride;
      if (i >= total) break;
      int k = 0;
      while (k + 1 < nnbr && i >= nb[k].off[dir] + nb[k].n[dir]) k++;
      st_sys_f64(&nb[k].dst[dir][(long long)par * nb[k].stride[dir] + (i - nb[k].off[dir])], val[j]);
    }
  }
  if (!last_block(&L->done[dir], mblk)) return;
  for (int k = threadIdx.x; k < nnbr; k += blockDim.x) {
    if (nb[k].n[dir] > 0) st_sys(nb[k].flag[dir], s);
    if (ack_p && nb[k].rn[0] > 0) st_sys(nb[k].ack[0], L->seq[0]);
  }
  if (threadIdx.x == 0) L->seq[dir] = s;
  if (MODE == 2) wait_exchange(nb, nnbr, F, dir, s, err);
}

// (2a) P: mailbox -> ghost slots, then the acknowledgements
__global__ __launch_bounds__(256) void k_peer_consume_p(const PeerNbr *__restrict__ nb, const int nnbr,
                                                         const PeerLocal *__restrict__ F, PeerCounters *__restrict__ L,
                                                         const double *__restrict__ mb, const int nrecv, double *__restrict__ v,
                                                         const int32_t *__restrict__ idx, const int first,
                                                         unsigned long long *err) {
  const unsigned long long s = L->seq[0];
  wait_exchange(nb, nnbr, F, 0, s, err);
  const double *src = mb + (size_t)(s & 1ull) * nrecv;
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < nrecv; i0 += 4 * stride) {
    double val[4];
    int d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * stride;
      val[j] = i < nrecv ? ld_sys_f64(&src[i]) : 0.0;
      d[j] = i < nrecv ? (idx ? idx[i] : first + i) : -1;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (d[j] >= 0) v[d[j]] = val[j];
  }
  if (!last_block(&L->cons[0], gridDim.x)) return;
  for (int k = threadIdx.x; k < nnbr; k += blockDim.x)
    if (nb[k].rn[0] > 0) st_sys(nb[k].ack[0], s);
}

// (2b) P^T: every owned dof that has sharers adds their contributions, neighbour by neighbour in plan order (fixed summation
// order: the result does not depend on arrival times).  FUSED (ParOperator::Mult): the result goes to y with ParOperator's
// essential rows fixed on the way (rap.cpp:222-233): y[i] = ess ? (x[i] | 0) : ly[i] + contributions, mask bit 1 = essential,
// bit 2 = the dof has sharers (its row of the contribution lists is found by bisection)
// MODE 2 (direct form): v = y already holds the local sums of the true dofs (the run gather wrote them, essential rows fixed);
// the contributions are added in place, essential rows are left alone
template <int MODE>
__global__ __launch_bounds__(256) void k_peer_consume_r(const PeerNbr *__restrict__ nb, const int nnbr,
                                                         const PeerLocal *__restrict__ F, PeerCounters *__restrict__ L,
                                                         const double *__restrict__ mb, const int nsend, double *__restrict__ v,
                                                         const int ndof, const int4 *__restrict__ rinfo,
                                                         const int32_t *__restrict__ rptr, const int32_t *__restrict__ rpos,
                                                         unsigned long long *err, const uint8_t *__restrict__ mask,
                                                         const double *__restrict__ x, const int diag_one, double *__restrict__ y,
                                                         const int n_true, const int sblk) {
  constexpr bool FUSED = MODE == 1;
  // Blocks [0, sblk) own the dofs with sharers (they wait for the messages, read the mailbox and acknowledge); FUSED: the
  // blocks behind them write every other true dof, one per thread, without waiting for anybody.
  if (FUSED && (int)blockIdx.x >= sblk) {
    const int i = ((int)blockIdx.x - sblk) * blockDim.x + threadIdx.x;
    if (i < n_true) {
      const uint8_t m = mask[i];
      if (!(m & 2)) y[i] = (m & 1) ? (diag_one ? x[i] : 0.0) : v[i];
    }
    return;
  }
  const unsigned long long s = L->seq[1];
  wait_exchange(nb, nnbr, F, 1, s, err);
  const double *src = mb + (size_t)(s & 1ull) * nsend;
  // rinfo[i] = {dof, position of its first contribution, of its second one or -1, first entry of the others in rpos or -1}:
  // one 16-byte load, then the vector entry and the first two contributions side by side (the usual case: one or two sharers)
  const int stride = sblk * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < ndof; i0 += 2 * stride) {
    int4 info[2];
    double sum[2], c0[2], c1[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j * stride;
      info[j] = i < ndof ? rinfo[i] : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      sum[j] = info[j].x >= 0 ? v[info[j].x] : 0.0;
      c0[j] = info[j].y >= 0 ? ld_sys_f64(&src[info[j].y]) : 0.0;
      c1[j] = info[j].z >= 0 ? ld_sys_f64(&src[info[j].z]) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j * stride, d = info[j].x;
      if (d < 0) break;
      double t = sum[j] + c0[j];
      if (info[j].z >= 0) t += c1[j];
      if (info[j].w >= 0)
        for (int a = info[j].w; a < rptr[i + 1]; a++) t += ld_sys_f64(&src[rpos[a]]);
      if (FUSED)
        y[d] = (mask[d] & 1) ? (diag_one ? x[d] : 0.0) : t;
      else if (MODE == 2) {
        if (!(mask[d] & 1)) v[d] = t;
      } else
        v[d] = t;
    }
  }
  if (!last_block(&L->cons[1], sblk)) return;
  for (int k = threadIdx.x; k < nnbr; k += blockDim.x)
    if (nb[k].rn[1] > 0) st_sys(nb[k].ack[1], s);
}

// (3) P^T of the direct form in ONE kernel: ghost rows (contiguous: the local apply wrote them to GhostOut) into the owners' mailboxes,
// flags, acknowledgement of this Mult's P messages -- then the same blocks wait for the neighbours' rows and add them to y in place
// (k_peer_send<0> + k_peer_consume_r<2> behind one launch: at the per-rank sizes of a strong-scaling run a launch costs what
// the exchange does).  Every block takes part in both phases; the exchange counter is advanced by the block that finishes last,
// after every block has read it.  A block spinning in the second phase waits for OTHER ranks' first phases only, which never wait.
// STEP: y is read only (this rank's partial sums of the interface dofs) and the completed sum is consumed by the smoother step
// `st` instead of being stored (Halo::Step)
template <bool STEP>
__global__ __launch_bounds__(256) void k_peer_restrict_direct_t(const PeerNbr *__restrict__ nb, const int nnbr, const PeerLocal *__restrict__ F,
                                                                 PeerCounters *__restrict__ L, const int total,
                                                                 const double *__restrict__ gout, const double *__restrict__ mb,
                                                                 const int nsend, double *__restrict__ y, const int ndof,
                                                                 const int4 *__restrict__ rinfo, const int32_t *__restrict__ rptr,
                                                                 const int32_t *__restrict__ rpos, unsigned long long *err,
                                                                 const uint8_t *__restrict__ mask, const Halo::Step st) {
  const int dir = 1;
  const unsigned long long s = L->seq[1] + 1ull;
  const int par = (int)(s & 1ull);
  const int stride = gridDim.x * blockDim.x;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += 4 * stride) {
    double val[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * stride;
      val[j] = i < total ? gout[i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * stride;
      if (i >= total) break;
      int k = 0;
      while (k + 1 < nnbr && i >= nb[k].off[dir] + nb[k].n[dir]) k++;
      st_sys_f64(&nb[k].dst[dir][(long long)par * nb[k].stride[dir] + (i - nb[k].off[dir])], val[j]);
    }
  }
  if (last_block(&L->done[1], gridDim.x)) {
    for (int k = threadIdx.x; k < nnbr; k += blockDim.x) {
      if (nb[k].n[dir] > 0) st_sys(nb[k].flag[dir], s);
      if (nb[k].rn[0] > 0) st_sys(nb[k].ack[0], L->seq[0]);  // the element kernel that read the P messages in place has finished
    }
  }
  wait_exchange(nb, nnbr, F, 1, s, err);
  const double *src = mb + (size_t)par * nsend;
  for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < ndof; i0 += 2 * stride) {
    int4 info[2];
    double sum[2], c0[2], c1[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j * stride;
      info[j] = i < ndof ? rinfo[i] : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      sum[j] = info[j].x >= 0 ? y[info[j].x] : 0.0;
      c0[j] = info[j].y >= 0 ? ld_sys_f64(&src[info[j].y]) : 0.0;
      c1[j] = info[j].z >= 0 ? ld_sys_f64(&src[info[j].z]) : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int i = i0 + j * stride, d = info[j].x;
      if (d < 0) break;
      double t = sum[j] + c0[j];
      if (info[j].z >= 0) t += c1[j];
      if (info[j].w >= 0)
        for (int a = info[j].w; a < rptr[i + 1]; a++) t += ld_sys_f64(&src[rpos[a]]);
      if (STEP) {
        const double tt = (mask[d] & 1) ? sum[j] : t;  // (essential rows: the local gather fixed them)
        const double rv = st.r0[d] - tt;
        if (st.mode == 2) {
          if (st.res) st.res[d] = rv;
          if (st.out) st.out[d] = st.sr * st.dinv[d] * rv;
        } else {
          const double e = st.ek[d];
          double dk = st.sr * st.dinv[d] * rv;
          dk += st.sd * (e - (st.ep ? st.ep[d] : 0.0));
          st.out[d] = (st.add ? st.out[d] : 0.0) + (e + dk);
        }
      } else if (!(mask[d] & 1)) {
        y[d] = t;
      }
    }
  }
  if (!last_block(&L->cons[1], gridDim.x)) return;
  for (int k = threadIdx.x; k < nnbr; k += blockDim.x)
    if (nb[k].rn[1] > 0) st_sys(nb[k].ack[1], s);
  if (threadIdx.x == 0) L->seq[1] = s;
}

// global sum: my values into everybody's slot of me, flags; then wait for everybody and add in rank order (the same result,
// bit for bit, on every rank).  Double-buffered: a rank can be one sum ahead of another, never two (it needs the other's flag
// of the sum in between).
struct ArChannel {
  size_t seq, flags, slots;  // byte offsets in every rank's arena
  int maxn;                  // doubles per rank and buffer
};
__global__ void k_ar_send(char *const *__restrict__ remote, const int me, const int size, char *__restrict__ mine,
                          const double *__restrict__ vals, const int n, const ArChannel ch) {
  unsigned long long *seq = reinterpret_cast<unsigned long long *>(mine + ch.seq);
  const unsigned long long s = *seq + 1ull;
  const size_t par = (size_t)(s & 1ull);
  for (int t = threadIdx.x; t < size * n; t += blockDim.x) {
    const int r = t / n, i = t - r * n;
    double *slots = reinterpret_cast<double *>(remote[r] + ch.slots);
    st_sys_f64(&slots[(par * Comm::kMaxRanks + me) * ch.maxn + i], vals[i]);
  }
  stores_performed();
  __syncthreads();
  for (int r = threadIdx.x; r < size; r += blockDim.x)
    st_sys(reinterpret_cast<unsigned long long *>(remote[r] + ch.flags) + me, s);
  if (threadIdx.x == 0) *seq = s;
}
__global__ void k_ar_sum(char *__restrict__ mine, const int size, const int n, double *__restrict__ out, const ArChannel ch,
                         unsigned long long *err) {
  const unsigned long long s = *reinterpret_cast<const unsigned long long *>(mine + ch.seq);
  const unsigned long long *flags = reinterpret_cast<const unsigned long long *>(mine + ch.flags);
  for (int r = threadIdx.x; r < size; r += blockDim.x) spin_ge(&flags[r], s, err);
  messages_arrived();
  __syncthreads();
  const double *slots = reinterpret_cast<const double *>(mine + ch.slots) + (size_t)(s & 1ull) * Comm::kMaxRanks * ch.maxn;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double sum = 0.0;
    for (int r = 0; r < size; r++) sum += ld_sys_f64(&slots[(size_t)r * ch.maxn + i]);
    out[i] = sum;
  }
}

}  // namespace

namespace {
bool g_fenced_host = false;      // what Comm::SetFenced / SetTimeout were last called with in this process (the current device's symbols
double g_timeout_host = 0.0;     // hold them; AllocArena re-applies them on the device of a new communicator); 0: default time-out
}  // namespace

void Comm::AllocArena() {
  const char *mb = std::getenv("PALACE_AMD_PEER_ARENA_MB");
  arena_bytes_ = (size_t)(mb ? std::max(8, atoi(mb)) : 256) << 20;
  PA_REQUIRE(size_ <= kMaxRanks, "too many ranks for the peer transport");
  // flags polled by one device while another one writes them: memory that is not cached incoherently
  const char *mode = std::getenv("PALACE_AMD_PEER_MEM");
  const std::string m = mode ? mode : "uncached";
  void *p = nullptr;
  hipError_t rc = hipErrorUnknown;
  if (m == "uncached") rc = hipExtMallocWithFlags(&p, arena_bytes_, hipDeviceMallocUncached);
  arena_uncached_ = rc == hipSuccess;  // (plain loads of other kernels may read the mailboxes in place: Halo::DirectOk)
  if (rc != hipSuccess && m != "plain") rc = hipExtMallocWithFlags(&p, arena_bytes_, hipDeviceMallocFinegrained);
  if (rc != hipSuccess) {
    (void)hipGetLastError();
    PA_HIP(hipMalloc(&p, arena_bytes_));
  }
  arena_ = static_cast<char *>(p);
  PA_HIP(hipMemset(arena_, 0, kOffDynamic));
  arena_used_ = kOffDynamic;
  halo_live_.assign((size_t)kMaxHalos, 0);
  PA_HIP(hipStreamCreateWithFlags(&setup_stream_, hipStreamNonBlocking));
  // the error word of the wait loops: host memory mapped into the device (written by a kernel only when a wait times out)
  void *he = nullptr;
  PA_HIP(hipHostMalloc(&he, 64, hipHostMallocMapped));
  h_err_ = static_cast<unsigned long long *>(he);
  *h_err_ = 0ull;
  // The two settings live in __device__ symbols, i.e. once per DEVICE, while SetTimeout / SetFenced are process-wide: a
  // communicator brought up on a device after they were called re-applies the process's values there (a process that drives
  // several devices would otherwise run them with different time-outs or fence modes while Fenced() reports one value).
  if (const char *t = std::getenv("PALACE_AMD_PEER_TIMEOUT_S")) SetTimeout(atof(t));
  else if (g_timeout_host > 0.0) SetTimeout(g_timeout_host);
  if (const char *f = std::getenv("PALACE_AMD_PEER_FENCE")) SetFenced(atoi(f) != 0);
  else if (g_fenced_host) SetFenced(true);
}

void Comm::SetTimeout(double seconds) {
  const long long ticks = (long long)(std::max(0.01, seconds) * 1.0e8);
  PA_HIP(hipDeviceSynchronize());
  PA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_spin_ticks), &ticks, sizeof(ticks)));
  g_timeout_host = seconds;
}
void Comm::SetFenced(bool on) {
  const int v = on ? 1 : 0;
  PA_HIP(hipDeviceSynchronize());
  PA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_peer_fence), &v, sizeof(v)));
  g_fenced_host = on;
}
bool Comm::Fenced() { return g_fenced_host; }

Comm::Comm(int rank, int size) : rank_(rank), size_(size) {
  PA_REQUIRE(size >= 1 && rank >= 0 && rank < size, "bad communicator arguments");
  AllocArena();
  if (size == 1) {
    remote_.assign(1, arena_), remote_ipc_.assign(1, 0);
    d_remote_ = pa::dev_upload(remote_.data(), 1);
  }
}

void Comm::PeerHandle(char *out) {
  if (!arena_) AllocArena();
  static_assert(sizeof(hipIpcMemHandle_t) == kPeerHandleBytes, "IPC handle size");
  hipIpcMemHandle_t h;
  PA_HIP(hipIpcGetMemHandle(&h, arena_));
  std::memcpy(out, &h, sizeof(h));
}

void Comm::PeerConnect(const char *handles) {
  PA_REQUIRE(arena_ && handles, "peer transport: no arena");
  PA_REQUIRE(remote_.empty() || size_ == 1, "peer transport is already connected");
  remote_.assign((size_t)size_, nullptr), remote_ipc_.assign((size_t)size_, 0);
  for (int r = 0; r < size_; r++) {
    if (r == rank_) {
      remote_[r] = arena_;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handles + (size_t)r * kPeerHandleBytes, sizeof(h));
    void *p = nullptr;
    const hipError_t rc = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (rc != hipSuccess) {  // all or nothing: a half-connected transport must not look ready
      (void)hipGetLastError();
      for (size_t q = 0; q < remote_.size(); q++)
        if (remote_ipc_[q]) (void)hipIpcCloseMemHandle(remote_[q]);
      remote_.clear(), remote_ipc_.clear();
      if (size_ == 1) remote_.assign(1, arena_), remote_ipc_.assign(1, 0);
      throw pa::Error(std::string("peer transport: cannot map the arena of rank ") + std::to_string(r) + ": " + hipGetErrorString(rc));
    }
    remote_[r] = static_cast<char *>(p), remote_ipc_[r] = 1;
  }
  if (d_remote_) (void)hipFree(d_remote_);
  d_remote_ = pa::dev_upload(remote_.data(), remote_.size());
}

void Comm::PeerDisconnect() {
  PA_REQUIRE(nccl_ || size_ == 1, "a communicator without RCCL cannot give up the peer transport");
  for (size_t r = 0; r < remote_.size(); r++)
    if (remote_ipc_[r]) (void)hipIpcCloseMemHandle(remote_[r]);
  remote_.clear(), remote_ipc_.clear();
  if (h_err_) *h_err_ = 0ull;  // (what the transport noted no longer concerns the RCCL path)
}

size_t Comm::PeerAlloc(size_t bytes) {
  bytes = (bytes + 255) & ~size_t(255);
  // a block a destroyed plan gave back (best fit, at most twice the size), else fresh space
  size_t best = arena_free_.size();
  for (size_t i = 0; i < arena_free_.size(); i++)
    if (arena_free_[i].second >= bytes && arena_free_[i].second <= 2 * bytes &&
        (best == arena_free_.size() || arena_free_[i].second < arena_free_[best].second))
      best = i;
  size_t off;
  if (best < arena_free_.size()) {
    off = arena_free_[best].first;
    arena_free_.erase(arena_free_.begin() + (long)best);
  } else {
    off = (arena_used_ + 255) & ~size_t(255);
    PA_REQUIRE(off + bytes <= arena_bytes_, "peer arena exhausted (PALACE_AMD_PEER_ARENA_MB)");
    arena_used_ = off + bytes;
  }
  PA_HIP(hipMemset(arena_ + off, 0, bytes));
  return off;
}
// A block given back may still receive a neighbour's last acknowledgement (its kernels run on its own clock): it waits in
// quarantine until the next set-up barrier, before which every rank drains its device (PeerSetup), and only then is used again.
void Comm::PeerFree(size_t off, size_t bytes) { arena_quarantine_.emplace_back(off, (bytes + 255) & ~size_t(255)); }

bool Comm::GraphSafe() const {
  if (size_ == 1) return true;
  const char *mode = std::getenv("PALACE_AMD_HALO");
  return PeerReady() && !(mode && std::string(mode) == "rccl") && !std::getenv("PALACE_AMD_PEER_NO_REDUCE");
}

void Comm::PeerCheck(hipStream_t s) {
  if (!h_err_) return;
  PA_HIP(hipStreamSynchronize(s));
  PeerCheckNow();
}
void Comm::PeerCheckNow() {
  if (!h_err_) return;
  if (*static_cast<volatile unsigned long long *>(h_err_)) {
    // (the word stays raised: the ranks are out of step from here on, every later check reports it again)
    throw pa::Error("peer transport: a wait for another rank's message timed out (a rank stopped or fell more than "
                    "PALACE_AMD_PEER_TIMEOUT_S behind, or the plans of two ranks do not match); results after this point are not valid");
  }
}

void Comm::PeerAllReduce(double *d_buf, int n, hipStream_t s, int channel) {
  PA_REQUIRE(channel >= 0 && channel < 2 && n <= kArMaxN[channel], "too many values for the peer all-reduce");
  const ArChannel ch{kOffArSeq[channel], kOffArFlags[channel], kOffArSlots[channel], kArMaxN[channel]};
  hipLaunchKernelGGL(k_ar_send, dim3(1), dim3(256), 0, s, d_remote_, rank_, size_, arena_, d_buf, n, ch);
  hipLaunchKernelGGL(k_ar_sum, dim3(1), dim3(256), 0, s, arena_, size_, n, d_buf, ch, h_err_);
  PA_HIP(hipGetLastError());
}

struct Halo::PeerPlan {
  PeerNbr *d_nbr = nullptr;
  PeerLocal *local = nullptr;       // in the arena
  PeerCounters *counters = nullptr;  // ordinary device memory
  double *mb[2] = {nullptr, nullptr};
  int nnbr = 0, n_rdof = 0;
  int4 *d_rinfo = nullptr;
  int32_t *d_rptr = nullptr, *d_rpos = nullptr;
  unsigned long long *d_err = nullptr;
  int slot = -1;                    // descriptor slot (given back with the plan)
  int ranks_on_device = 1;          // Comm::RanksOnMyDevice at set-up: the share of the device a resident grid may take
  size_t off[3] = {0, 0, 0}, bytes[3] = {0, 0, 0};  // my arena blocks: mailboxes of P / P^T, flags
};

void Halo::PeerSetup(const int32_t *send_idx) {
  Comm &c = *comm_;
  const int nn = (int)nbr_.size();
  // Plan ids count the plans made on this communicator (collective: the same on every rank).  A rank keeps the plan's
  // descriptor in any free slot of its table (slots and arena blocks of destroyed plans are used again -- each rank on its own
  // clock: destructors run when the host language gets to them) and tells the others which one in the set-up barrier; the
  // descriptor carries the plan id.  Everything that can fail locally is decided BEFORE the barrier, and a rank that fails
  // still takes part in it (reporting slot -1), so that the others fail on that instead of waiting for it.
  const int id = c.next_halo_++;
  int slot = 0;
  while (slot < Comm::kMaxHalos && c.halo_live_[(size_t)slot]) slot++;
  PA_HIP(hipDeviceSynchronize());  // (my stores into other ranks' arenas -- acknowledgements of destroyed plans -- are performed)
  std::string fail;
  if (nn > Comm::kMaxNbr) fail = "too many neighbours for the peer transport";
  if (slot == Comm::kMaxHalos) fail = "too many live halo plans for the peer transport", slot = -1;
  const size_t want[3] = {sizeof(double) * 2 * (size_t)std::max(1, nrecv_), sizeof(double) * 2 * (size_t)std::max(1, nsend_),
                          sizeof(PeerLocal)};
  HaloDesc d;
  std::memset(&d, 0, sizeof(d));
  auto *pp = new PeerPlan;
  if (fail.empty()) {
    try {
      for (int b = 0; b < 3; b++) pp->off[b] = c.PeerAlloc(want[b]), pp->bytes[b] = want[b];
    } catch (const pa::Error &e) {
      fail = e.what();
      for (int b = 0; b < 3; b++)
        if (pp->bytes[b]) c.PeerFree(pp->off[b], pp->bytes[b]), pp->bytes[b] = 0;
    }
  }
  if (fail.empty()) {
    d.nnbr = nn, d.nrecv = nrecv_, d.nsend = nsend_;
    for (int k = 0; k < nn; k++) d.nbr[k] = nbr_[k];
    for (int k = 0; k <= nn; k++) d.recv_off[k] = recv_off_[k], d.send_off[k] = send_off_[k];
    d.off_mb[0] = pp->off[0], d.off_mb[1] = pp->off[1], d.off_local = pp->off[2];
    d.ready = kDescMagic + (unsigned long long)id;
  }
  if (slot >= 0) {
    // On the set-up stream, and waited for: a synchronous copy from pageable memory may return when the descriptor has reached
    // the staging buffer, not the arena -- a neighbour released by the barrier below (another stream) could then read the
    // descriptor of the plan that held this slot before (seen once in 1 100 plan creations of the churn test, round 5).
    PA_HIP(hipMemcpyAsync(c.arena_ + kOffDesc + sizeof(HaloDesc) * (size_t)slot, &d, sizeof(d), hipMemcpyHostToDevice, c.setup_stream_));
    PA_HIP(hipStreamSynchronize(c.setup_stream_));
  }
  std::vector<double> slots;
  try {
    slots = c.SetupGather(fail.empty() ? (double)slot : -1.0, c.setup_stream_);  // every rank has published plan `id`
  } catch (...) {
    delete pp;
    throw;
  }
  // every rank has drained its device and passed the barrier: what was given back before it can be used again
  c.arena_free_.insert(c.arena_free_.end(), c.arena_quarantine_.begin(), c.arena_quarantine_.end());
  c.arena_quarantine_.clear();
  if (!fail.empty()) {
    delete pp;
    throw pa::Error(fail);
  }
  c.halo_live_[(size_t)slot] = 1;
  pp->slot = slot;
  try {
    pp->ranks_on_device = c.RanksOnMyDevice(c.setup_stream_);  // (collective at its first call: every rank is here)
  } catch (...) {
    pp->ranks_on_device = c.Size();
  }
  peer_ = pp;
  try {  // (a throwing constructor runs no destructor: the slot and the blocks go back here)
  pp->nnbr = nn;
  pp->local = reinterpret_cast<PeerLocal *>(c.arena_ + d.off_local);
  pp->counters = pa::dev_alloc<PeerCounters>(1);
  PA_HIP(hipMemset(pp->counters, 0, sizeof(PeerCounters)));
  pp->mb[0] = reinterpret_cast<double *>(c.arena_ + d.off_mb[0]);
  pp->mb[1] = reinterpret_cast<double *>(c.arena_ + d.off_mb[1]);
  pp->d_err = c.h_err_;
  std::vector<PeerNbr> nb((size_t)nn);
  for (int k = 0; k < nn; k++) {
    const int r = nbr_[k];
    HaloDesc rd;
    const int rslot = (int)slots[(size_t)r];
    PA_REQUIRE(rslot >= 0 && rslot < Comm::kMaxHalos, "peer transport: a neighbour could not set up this halo plan");
    PA_HIP(hipMemcpy(&rd, c.remote_[r] + kOffDesc + sizeof(HaloDesc) * (size_t)rslot, sizeof(rd), hipMemcpyDeviceToHost));
    PA_REQUIRE(rd.ready == kDescMagic + (unsigned long long)id,
               "peer transport: a neighbour has not published this halo plan (it failed to set it up, or the ranks create "
               "their plans in different orders)");
    int j = 0;
    while (j < rd.nnbr && rd.nbr[j] != c.rank_) j++;
    const int ns = send_off_[k + 1] - send_off_[k], nr = recv_off_[k + 1] - recv_off_[k];
    PA_REQUIRE(j < rd.nnbr && rd.recv_off[j + 1] - rd.recv_off[j] == ns && rd.send_off[j + 1] - rd.send_off[j] == nr,
               "halo plans of two ranks do not match");
    PeerNbr &q = nb[k];
    char *rb = c.remote_[r];
    PeerLocal *rl = reinterpret_cast<PeerLocal *>(rb + rd.off_local);
    q.dst[0] = reinterpret_cast<double *>(rb + rd.off_mb[0]) + rd.recv_off[j], q.stride[0] = rd.nrecv;
    q.dst[1] = reinterpret_cast<double *>(rb + rd.off_mb[1]) + rd.send_off[j], q.stride[1] = rd.nsend;
    for (int dir = 0; dir < 2; dir++) q.flag[dir] = &rl->flag[dir][j], q.ack[dir] = &rl->ack[dir][j];
    q.off[0] = send_off_[k], q.n[0] = ns, q.rn[0] = nr;
    q.off[1] = recv_off_[k], q.n[1] = nr, q.rn[1] = ns;
  }
  pp->d_nbr = pa::dev_upload(nb.data(), nb.size());
  // P^T: the owned dofs with sharers and the positions of their contributions in the mailbox, neighbour-major
  {
    std::vector<int32_t> order((size_t)nsend_);
    for (int i = 0; i < nsend_; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return send_idx[a] < send_idx[b]; });
    std::vector<int32_t> rdof, rptr(1, 0), rpos;
    for (int i = 0; i < nsend_; i++) {
      if (i == 0 || send_idx[order[i]] != send_idx[order[i - 1]]) {
        if (i) rptr.push_back((int32_t)rpos.size());
        rdof.push_back(send_idx[order[i]]);
      }
      rpos.push_back(order[i]);
    }
    if (nsend_) rptr.push_back((int32_t)rpos.size());
    pp->n_rdof = (int)rdof.size();
    shared_owned_.assign(rdof.begin(), rdof.end());
    std::vector<int4> rinfo(rdof.size());
    for (size_t i = 0; i < rdof.size(); i++) {
      const int b = rptr[i], e = rptr[i + 1];
      rinfo[i] = make_int4(rdof[i], rpos[b], e - b > 1 ? rpos[b + 1] : -1, e - b > 2 ? b + 2 : -1);
    }
    pp->d_rinfo = pa::dev_upload(rinfo.data(), rinfo.size());
    pp->d_rptr = pa::dev_upload(rptr.data(), rptr.size());
    pp->d_rpos = pa::dev_upload(rpos.data(), rpos.size());
  }
  } catch (...) {
    FreePeer();
    throw;
  }
}

void Halo::FreePeer() {
  if (!peer_) return;
  // (hipFree waits for the device: no exchange of this plan is in flight on this rank when its blocks go back to the arena;
  // the neighbours stop using their views of them with their own plan objects -- plans are created and destroyed collectively)
  (void)hipFree(peer_->counters), (void)hipFree(peer_->d_nbr), (void)hipFree(peer_->d_rinfo), (void)hipFree(peer_->d_rptr), (void)hipFree(peer_->d_rpos);
  for (int b = 0; b < 3; b++)
    if (peer_->bytes[b]) comm_->PeerFree(peer_->off[b], peer_->bytes[b]);
  if (peer_->slot >= 0) comm_->halo_live_[(size_t)peer_->slot] = 0;
  delete peer_;
  peer_ = nullptr;
}

// dir 0: P (owners -> ghosts), 1: P^T (ghosts -> owners, added)
void Halo::PeerExchange(int dir, double *d_v, hipStream_t s) const {
  const PeerPlan &p = *peer_;
  const int total = dir == 0 ? nsend_ : nrecv_;
  const int32_t *sidx = dir == 0 ? d_send_idx_ : (recv_first_ >= 0 ? nullptr : d_recv_idx_);
  const int mb = mail_blocks(total);
  hipLaunchKernelGGL(k_peer_send<0>, dim3(mb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.counters, dir, total, d_v, sidx,
                     dir == 0 ? 0 : recv_first_, nullptr, nullptr, 0, mb, nullptr, nullptr, 0);
  if (dir == 0) {
    hipLaunchKernelGGL(k_peer_consume_p, dim3(mail_blocks(nrecv_)), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local, p.counters, p.mb[0],
                       nrecv_, d_v, recv_first_ >= 0 ? nullptr : d_recv_idx_, recv_first_, p.d_err);
  } else {
    const int sb = sum_blocks(p.n_rdof);
    hipLaunchKernelGGL(k_peer_consume_r<0>, dim3(sb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local, p.counters, p.mb[1], nsend_,
                       d_v, p.n_rdof, p.d_rinfo, p.d_rptr, p.d_rpos, p.d_err, nullptr, nullptr, 0, nullptr, 0, sb);
  }
  PA_HIP(hipGetLastError());
}

// The two exchanges of ParOperator::Mult with the vector copies around them folded in (peer transport only):
//   lx = P (x with the essential entries zeroed)                 one send kernel + the ghost unpack
//   y  = P^T ly with the essential rows set to x | 0             one send kernel + one kernel for everything else
// mask [n_true]: bit 1 essential, bit 2 owned dof with sharers (SharedOwnedDofs()).
void Halo::ProlongateFused(const double *d_x, const uint8_t *d_mask, int n_true, double *d_lx, hipStream_t s) const {
  const PeerPlan &p = *peer_;
  const int mb = mail_blocks(nsend_);
  hipLaunchKernelGGL(k_peer_send<1>, dim3(mb + (n_true + 255) / 256), dim3(256), 0, s, p.d_nbr, p.nnbr, p.counters, 0, nsend_,
                     d_x, d_send_idx_, 0, d_mask, d_lx, n_true, mb, nullptr, nullptr, 0);
  hipLaunchKernelGGL(k_peer_consume_p, dim3(mail_blocks(nrecv_)), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local, p.counters, p.mb[0],
                     nrecv_, d_lx, recv_first_ >= 0 ? nullptr : d_recv_idx_, recv_first_, p.d_err);
  PA_HIP(hipGetLastError());
}
void Halo::RestrictAddFused(const double *d_ly, const double *d_x, const uint8_t *d_mask, bool diag_one, int n_true, double *d_y,
                            hipStream_t s) const {
  const PeerPlan &p = *peer_;
  const int mb = mail_blocks(nrecv_), sb = sum_blocks(p.n_rdof);
  hipLaunchKernelGGL(k_peer_send<0>, dim3(mb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.counters, 1, nrecv_, d_ly,
                     recv_first_ >= 0 ? nullptr : d_recv_idx_, recv_first_, nullptr, nullptr, 0, mb, nullptr, nullptr, 0);
  hipLaunchKernelGGL(k_peer_consume_r<1>, dim3(sb + (n_true + 255) / 256), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local,
                     p.counters, p.mb[1], nsend_, const_cast<double *>(d_ly), p.n_rdof, p.d_rinfo, p.d_rptr, p.d_rpos, p.d_err,
                     d_mask, d_x, diag_one ? 1 : 0, d_y, n_true, sb);
  PA_HIP(hipGetLastError());
}

// The direct form of ParOperator::Mult (no L-vectors; local operators with a split-vector apply, pa_op_mult_split):
//   SendDirect:        masked x of the owned dofs with sharers -> the neighbours' mailboxes; returns when theirs have arrived
//   GhostIn*:          where the element kernel then reads the ghosts: the mailbox itself (two buffers, the parity of the
//                      device-resident exchange counter says which -- the kernels are recordable)
//   RestrictAddDirect: ghost rows (GhostOut) -> the owners' mailboxes, acknowledgement of the P messages; then every owned dof
//                      with sharers adds their contributions to y in place (essential rows are left as the gather fixed them)
bool Halo::DirectOk(int n_true, int n_local) const {
  // (the element kernel reads the mailbox with plain loads: only from an uncached arena, where no stale line can be hit)
  return peer_ != nullptr && comm_->arena_uncached_ && nrecv_ == n_local - n_true && (nrecv_ == 0 || recv_first_ == n_true);
}
const double *Halo::GhostIn(int buffer) const { return peer_->mb[0] + (size_t)buffer * nrecv_; }
const unsigned long long *Halo::GhostInSelector() const { return &peer_->counters->seq[0]; }
double *Halo::GhostOut() const {
  if (!d_ghost_out_) d_ghost_out_ = pa::dev_alloc<double>((size_t)std::max(1, nrecv_));
  return d_ghost_out_;
}
void Halo::SendDirect(const double *d_x, const uint8_t *d_mask, hipStream_t s) const {
  const PeerPlan &p = *peer_;
  const int mb = mail_blocks(nsend_);
  hipLaunchKernelGGL(k_peer_send<2>, dim3(mb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.counters, 0, nsend_, d_x, d_send_idx_, 0, d_mask,
                     nullptr, 0, mb, p.local, p.d_err, 0);
  PA_HIP(hipGetLastError());
}
namespace {
bool halo_merged() {
  static const bool merged = !(std::getenv("PALACE_AMD_HALO_MERGED") && std::getenv("PALACE_AMD_HALO_MERGED")[0] == '0');
  return merged;
}
int restrict_resident_blocks() {
  static const int resident = [] {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop{};
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_peer_restrict_direct_t<true>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1)
      prop.multiProcessorCount = 32;
    return per_cu * prop.multiProcessorCount;
  }();
  return resident;
}
}  // namespace
bool Halo::StepOk() const { return peer_ != nullptr && halo_merged(); }
void Halo::RestrictAddDirectStep(const uint8_t *d_mask, const double *d_t_iface, const Step &st, hipStream_t s) const {
  PA_REQUIRE(StepOk() && d_mask && d_t_iface && st.r0 && (st.mode == 1 ? (st.dinv && st.ek && st.out) : (st.mode == 2 && (st.res || st.out))),
             "halo step: peer transport in its merged form and a complete step expected");
  const PeerPlan &p = *peer_;
  const int mb = mail_blocks(nrecv_), sb = sum_blocks(p.n_rdof);
  const int cap = std::max(1, restrict_resident_blocks() / std::max(1, p.ranks_on_device));
  hipLaunchKernelGGL(k_peer_restrict_direct_t<true>, dim3(std::min(cap, std::max(mb, sb))), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local,
                     p.counters, nrecv_, GhostOut(), p.mb[1], nsend_, const_cast<double *>(d_t_iface), p.n_rdof, p.d_rinfo, p.d_rptr, p.d_rpos,
                     p.d_err, d_mask, st);
  PA_HIP(hipGetLastError());
}
void Halo::RestrictAddDirect(const uint8_t *d_mask, double *d_y, hipStream_t s) const {
  const PeerPlan &p = *peer_;
  const int mb = mail_blocks(nrecv_), sb = sum_blocks(p.n_rdof);
  static const bool merged = !(std::getenv("PALACE_AMD_HALO_MERGED") && std::getenv("PALACE_AMD_HALO_MERGED")[0] == '0');
  if (merged) {  // one launch for both halves (PALACE_AMD_HALO_MERGED=0: the two kernels of round 3)
    // Every block of this kernel takes part in both phases: a block spinning in the second one waits for the FIRST phase of other
    // ranks' grids, so each grid has to be resident as a whole -- also when the ranks share one device (rehearsals, CU-masked or
    // partitioned devices).  Both loops are grid-stride: clamp the grid to this rank's share of what the device holds at once.
    static const int resident = [] {
      int per_cu = 0, dev = 0;
      hipDeviceProp_t prop{};
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_peer_restrict_direct_t<true>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1)
        prop.multiProcessorCount = 32;
      return per_cu * prop.multiProcessorCount;
    }();
    // (this rank's share: the ranks on THIS device -- one per GPU on the target node, where nothing has to be shared; a
    // rehearsal or a partitioned device puts several on one.  Determined at the plan's set-up, PeerSetup.)
    const int cap = std::max(1, resident / std::max(1, p.ranks_on_device));
    hipLaunchKernelGGL(k_peer_restrict_direct_t<false>, dim3(std::min(cap, std::max(mb, sb))), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local, p.counters, nrecv_,
                       GhostOut(), p.mb[1], nsend_, d_y, p.n_rdof, p.d_rinfo, p.d_rptr, p.d_rpos, p.d_err, d_mask, Step{});
    PA_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(k_peer_send<0>, dim3(mb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.counters, 1, nrecv_, GhostOut(), nullptr, 0,
                     nullptr, nullptr, 0, mb, nullptr, nullptr, 1);
  hipLaunchKernelGGL(k_peer_consume_r<2>, dim3(sb), dim3(256), 0, s, p.d_nbr, p.nnbr, p.local, p.counters, p.mb[1], nsend_, d_y,
                     p.n_rdof, p.d_rinfo, p.d_rptr, p.d_rpos, p.d_err, d_mask, nullptr, 0, nullptr, 0, sb);
  PA_HIP(hipGetLastError());
}

// ---- stress test of the transport (Comm::StressRing) ----------------------------------------------------------------------
namespace {

// payloads: small integers (exact in double, exact sums), different for every rank, round, entry and purpose
__device__ __forceinline__ double stress_val(const int rank, const unsigned long long round, const int i, const unsigned salt) {
  const unsigned h = (unsigned)rank * 1315423911u + (unsigned)round * 2654435761u + (unsigned)i * 97u + salt * 40503u;
  return (double)((h ^ (h >> 13)) & 0xfffffu);
}
// lx[0, n) = owned payload of this round, ghosts poisoned
__global__ void k_stress_fill(double *lx, const int n, const int rank, const unsigned long long *round) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lx[i] = stress_val(rank, *round, i, 1u), lx[n + i] = -1.0;
}
// after P: ghost i (read from `g`, the L-vector tail or the mailbox buffer the exchange counter names) = the left neighbour's
// owned value; then the ghost rows of P^T are written to `gout`
__global__ void k_stress_check_p(const double *g0, const double *g1, const unsigned long long *sel, double *gout, const int n,
                                 const int rank, const int left, const unsigned long long *round, unsigned long long *fail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *g = (sel && (*sel & 1ull)) ? g1 : g0;
  if (g[i] != stress_val(left, *round, i, 1u)) atomicAdd(fail, 1ull);
  gout[i] = stress_val(rank, *round, i, 2u);
}
// after P^T: owned i = its own value + the right neighbour's ghost row
__global__ void k_stress_check_r(const double *y, const int n, const int rank, const int right, const unsigned long long *round,
                                 unsigned long long *fail, double *red, const int nred) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && y[i] != stress_val(rank, *round, i, 1u) + stress_val(right, *round, i, 2u)) atomicAdd(fail, 1ull);
  if (i < nred) red[i] = stress_val(rank, *round, i, 3u);
}
__global__ void k_stress_check_sum(const double *red, const int nred, const int size, unsigned long long *round,
                                   unsigned long long *fail) {
  const int i = threadIdx.x;
  if (i < nred) {
    double want = 0.0;
    for (int r = 0; r < size; r++) want += stress_val(r, *round, i, 3u);
    if (red[i] != want) atomicAdd(fail, 1ull);
  }
  __syncthreads();
  if (i == 0) *round += 1ull;
}

}  // namespace

long long Comm::StressRing(const Halo &ring, int n, int rounds, bool direct, bool graph, hipStream_t s) {
  PA_REQUIRE(size_ > 1 && PeerReady() && ring.UsesPeerTransport(), "stress test: needs the connected peer transport");
  PA_REQUIRE(!direct || ring.DirectOk(n, 2 * n), "stress test: the plan has no direct form");
  const int left = (rank_ + size_ - 1) % size_, right = (rank_ + 1) % size_, nred = 5, nb = (n + 255) / 256;
  double *lx = pa::dev_alloc<double>((size_t)2 * n), *red = pa::dev_alloc<double>(8);
  unsigned long long *st = pa::dev_alloc<unsigned long long>(2);  // {round, failures}
  uint8_t *mask = pa::dev_alloc<uint8_t>((size_t)n);
  PA_HIP(hipMemsetAsync(st, 0, 2 * sizeof(unsigned long long), s));
  PA_HIP(hipMemsetAsync(mask, 0, (size_t)n, s));
  auto round = [&] {
    hipLaunchKernelGGL(k_stress_fill, dim3(nb), dim3(256), 0, s, lx, n, rank_, st);
    if (direct) {
      // ParOperator::Mult's direct form: ghosts are read from the mailbox in place, ghost rows leave from GhostOut()
      ring.SendDirect(lx, mask, s);
      hipLaunchKernelGGL(k_stress_check_p, dim3(nb), dim3(256), 0, s, ring.GhostIn(0), ring.GhostIn(1), ring.GhostInSelector(),
                         ring.GhostOut(), n, rank_, left, st, st + 1);
      ring.RestrictAddDirect(mask, lx, s);
    } else {
      ring.Prolongate(lx, s);
      hipLaunchKernelGGL(k_stress_check_p, dim3(nb), dim3(256), 0, s, lx + n, lx + n, (const unsigned long long *)nullptr, lx + n,
                         n, rank_, left, st, st + 1);
      ring.RestrictAdd(lx, s);
    }
    hipLaunchKernelGGL(k_stress_check_r, dim3(nb), dim3(256), 0, s, lx, n, rank_, right, st, st + 1, red, nred);
    PeerAllReduce(red, nred, s);
    hipLaunchKernelGGL(k_stress_check_sum, dim3(1), dim3(64), 0, s, red, nred, size_, st, st + 1);
    PA_HIP(hipGetLastError());
  };
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  int done = 0;
  if (graph && rounds > 2) {
    round(), round(), done = 2;  // (both mailbox buffers have been used once: what a recorded solver iteration starts from)
    PA_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    round();
    PA_HIP(hipStreamEndCapture(s, &g));
    PA_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  }
  for (; done < rounds; done++) {
    if (ge)
      PA_HIP(hipGraphLaunch(ge, s));
    else
      round();
    if ((done & 1023) == 1023) {  // (bounded queue depth; a time-out surfaces here instead of after minutes of them)
      PA_HIP(hipStreamSynchronize(s));
      PeerCheckNow();
    }
  }
  unsigned long long h[2] = {0, 0};
  PA_HIP(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, s));
  PA_HIP(hipStreamSynchronize(s));
  if (ge) (void)hipGraphExecDestroy(ge);
  if (g) (void)hipGraphDestroy(g);
  (void)hipFree(lx), (void)hipFree(red), (void)hipFree(st), (void)hipFree(mask);
  PeerCheckNow();
  PA_REQUIRE((int)h[0] == rounds, "stress test: round counter out of step");
  return (long long)h[1];
}

}  // namespace palace
