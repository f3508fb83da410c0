// Communication for the element-partitioned operator: RCCL over xGMI, one process per GPU.
//
// Replaces on the hot path: the conforming prolongation P / P^T that Palace gets from MFEM's
// GroupCommunicator (palace/linalg/rap.cpp:212,216,222,363-373; MPI_Isend/Irecv per neighbour) and
// Mpi::GlobalSum (palace/utils/communication.hpp:249-252,270-273) behind every linalg::Dot.
//
// Data layout: a rank's local (L-) vector holds its true (owned) dofs first, [0, n_true), then the
// shared dofs owned by other ranks ("ghosts"), [n_true, n_local).  P copies owner values into the
// ghost slots of the sharers; P^T adds ghost contributions back onto the owners.  Both are sparse
// neighbour exchanges: every pair of GPUs has its own xGMI link, so all neighbours are served
// concurrently inside one ncclGroup (no ring, no staging through the host).
//
// Peer transport (default where it can be set up): every rank owns one device arena that the other ranks of the node map
// into their address space (hipIpcGetMemHandle / hipIpcOpenMemHandle).  A halo exchange is
// then three plain kernels on the caller's stream -- (1) pack my pieces straight into the neighbours' mailboxes (stores that
// travel over xGMI) and raise their flags, (2) one small block that waits for my flags, (3) unpack -- and a global sum is two:
// no RCCL call, nothing the host has to order, so the sequence can be recorded in a HIP graph like the one-rank solvers'.
// Mailboxes are double-buffered and acknowledged; sequence numbers live in device memory (a replayed graph advances them);
// every wait loop gives up after a few seconds and raises an error word instead of hanging the GPU.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include <condition_variable>
#include <mutex>

namespace palace {

// In-process stand-in for the RCCL communicator: `size` ranks are THREADS of one process on one GPU, each with its own Context and
// stream; collectives rendezvous on a host barrier and move data with device copies.  Exists so that every multi-rank code path
// (halo P / P^T around the operators, transfers and gradients with ghosts, global dots inside the Krylov loops, the device-resident
// PCG scalars) can be run and checked against the serial result on the one-GPU boxes (tests/test_multirank_local_gpu.py); slow by
// construction (host synchronisation in every collective) and not a product path.
class LocalGroup {
  friend class Comm;
  friend class Halo;
  const int size_;
  std::mutex m_;
  std::condition_variable cv_;
  int waiting_ = 0;
  long generation_ = 0;
  bool aborted_ = false;
  int timeout_s_ = 300;
  std::vector<double> slots_;  // [size][kMaxValues] staging of AllReduceSum
  struct Box {
    const double *buf = nullptr;  // published send buffer of a rank ...
    const int *nbr = nullptr;     // ... its neighbour list and the offsets of their pieces
    const int *off = nullptr;
    int nnbr = 0;
  };
  std::vector<Box> box_;

public:
  static constexpr int kMaxValues = 512;
  explicit LocalGroup(int size) : size_(size), slots_((size_t)size * kMaxValues), box_((size_t)size) {}
  int Size() const { return size_; }
  void Arrive();  // barrier over the ranks (threads); throws once the group is aborted or a rank is 300 s late
  void Abort();   // a failing rank releases the others (they throw out of their barrier)
};

class Comm {
  int rank_ = 0, size_ = 1;
  void *nccl_ = nullptr;  // ncclComm_t
  LocalGroup *local_ = nullptr;
  friend class Halo;
  // ---- peer transport ----
  char *arena_ = nullptr;              // my arena (device memory the other ranks map)
  size_t arena_bytes_ = 0, arena_used_ = 0;
  bool arena_uncached_ = false;
  std::vector<char *> remote_;         // [size] arena of rank r as mapped here (remote_[rank_] == arena_); empty: not connected
  std::vector<char> remote_ipc_;       // [size] 1: opened with hipIpcOpenMemHandle (to be closed)
  int next_halo_ = 0;                  // halo plans made so far (collective, monotonic: plan id)
  int ranks_on_device_ = 0;            // RanksOnMyDevice (0: not determined yet)
  std::vector<char> halo_live_;        // [kMaxHalos] descriptor slot in use
  std::vector<std::pair<size_t, size_t>> arena_free_;  // {offset, bytes} blocks given back by destroyed plans
  std::vector<std::pair<size_t, size_t>> arena_quarantine_;  // ... since the last set-up barrier (not yet reusable)
  unsigned long long *h_err_ = nullptr;  // error word of the wait loops: page-locked host memory the kernels write on a
                                         // time-out, so that every host synchronisation point can look at it for free
  char **d_remote_ = nullptr;          // device copy of remote_
  double *d_one_ = nullptr;            // Barrier's operand (per communicator)
  hipStream_t setup_stream_ = nullptr; // the transport's own stream for set-up barriers (never the shared null stream:
                                       // the rank threads of an in-process group would queue behind each other's waits)
  void AllocArena();
  void PeerAllReduce(double *d_buf, int n, hipStream_t s, int channel = 0);
  void PeerFree(size_t off, size_t bytes);

public:
  static constexpr int kUniqueIdBytes = 128;
  static constexpr int kPeerHandleBytes = 64;  // sizeof(hipIpcMemHandle_t)
  // (kMaxNbr = kMaxRanks: the gather plan of a replicated coarse solve names every other rank as a neighbour)
  static constexpr int kMaxRanks = 64, kMaxReduce = 4096, kMaxHalos = 512, kMaxNbr = 64;
  static constexpr int kMaxReduceSetup = kMaxRanks;  // values of the set-up channel (barriers / gathers of PeerSetup: own counters and slots)
  static void GetUniqueId(char *out);
  Comm(int rank, int size, const char *unique_id);
  Comm(int rank, LocalGroup &group);  // rank of an in-process group (see LocalGroup)
  // Communicator without RCCL: the peer transport carries the halo exchanges and the global sums.  The caller gathers
  // the arena handles of all ranks (PeerHandle; with MPI in Palace, torch.distributed in the tests) and hands them to
  // PeerConnect -- also what lets two processes share ONE GPU, which RCCL refuses.
  Comm(int rank, int size);
  ~Comm();
  int Rank() const { return rank_; }
  int Size() const { return size_; }
  void PeerHandle(char *out64);
  void PeerConnect(const char *handles);  // [size][kPeerHandleBytes]
  bool PeerReady() const { return !remote_.empty(); }
  void PeerDisconnect();  // back to RCCL (a failed self-test of the transport on this machine)
  // halo exchanges and global sums are plain kernels on the caller's stream: sequences containing them can be recorded
  bool GraphSafe() const;
  // raises if a wait loop of the peer transport has timed out since the last check (waits for the stream)
  void PeerCheck(hipStream_t s);
  // the same without waiting: for code that has just synchronised the stream anyway (every Dot / Sum / solver statistic
  // read-back calls it, so a lost message surfaces as an error at the next host synchronisation point, not as a wrong result)
  void PeerCheckNow();
  // conservative ordering: system-scope release / acquire fences around the flag stores and waits instead of relaxed atomics +
  // s_waitcnt (an L2 write-back per exchange kernel: slower; the fall-back tier if the relaxed protocol fails its self-test
  // on a machine).  Process-wide.
  static void SetFenced(bool on);
  static bool Fenced();
  // time limit of the device-side waits (default 60 s, PALACE_AMD_PEER_TIMEOUT_S); process-wide
  static void SetTimeout(double seconds);
  // Stress test of the transport on a ring plan of this communicator (`ring`: Halo over a vector of 2 n entries, owned [0, n),
  // ghosts [n, 2 n) owned by the left neighbour and sent to the right one): `rounds` rounds of P, P^T (L-vector or direct
  // form) and a global sum with payloads that change every round, every value verified ON THE DEVICE against its closed form;
  // graph: the round is recorded once and replayed.  Returns the number of wrong values seen by this rank (time-outs throw).
  long long StressRing(const class Halo &ring, int n, int rounds, bool direct, bool graph, hipStream_t s);
  // offset of a fresh piece of my arena
  size_t PeerAlloc(size_t bytes);
  char *PeerBase(int r) const { return remote_[r]; }
  // in-place sum over ranks of n doubles in device memory (Mpi::GlobalSum)
  void AllReduceSum(double *d_buf, int n, hipStream_t s);
  void Barrier(hipStream_t s);
  // barrier that also tells every rank one number of every other rank (set-up channel of the peer transport; size > 1)
  std::vector<double> SetupGather(double mine, hipStream_t s);
  // how many ranks of this communicator run on the same device as this one (PCI identity exchanged once over the set-up
  // channel; in-process groups share one device by construction).  Collective at its first call (Halo set-up).
  int RanksOnMyDevice(hipStream_t s);
  // Set-up time all-gather-v of host arrays over the peer transport (PeerReady): every rank stages its piece, a chunk at a time,
  // in a block of its own arena and reads the other ranks' chunks through their mappings -- each value crosses once per reader,
  // where the sum of zero-padded global arrays moved size times as much through 4096-value messages.  Returns the pieces in rank
  // order; offsets [size + 1] on request.  Collective.
  std::vector<double> AllGatherVHost(const std::vector<double> &mine, std::vector<long long> *offsets = nullptr);
};

// A smoother step consumed by the halo kernel (Halo::RestrictAddDirectStep)
struct HaloStep {
  int mode;
  double sd, sr;
  const double *dinv, *r0, *ek, *ep;
  double *out;
  int add;
  double *res;
};

// The conforming prolongation of one finite element space (one multigrid level): which owned dofs
// go to which neighbour and which ghost slots are filled by whom.
class Halo {
  Comm *comm_;
  std::vector<int> nbr_;                  // neighbour ranks
  std::vector<int> send_off_, recv_off_;  // [nnbr + 1] offsets into the index lists / buffers
  int32_t *d_send_idx_ = nullptr;  // owned dofs this rank sends in P (and receives-into in P^T)
  int32_t *d_recv_idx_ = nullptr;  // ghost slots this rank receives in P (and sends in P^T)
  double *d_sendbuf_ = nullptr, *d_recvbuf_ = nullptr;
  int recv_first_ = -1;  // >= 0: the ghosts are the contiguous range [recv_first_, recv_first_ + nrecv_) of the local vector in
                         // receive order (ghosts last: the usual numbering) -- received into / sent from it in place
  int nsend_ = 0, nrecv_ = 0;
  int send_min_ = 0, send_max_ = -1, recv_min_ = 0, recv_max_ = -1;  // index ranges of the plan (Validate)
  // in-process group: pieces [off[k], off[k + 1]) of `sendbase` go to neighbour k, pieces of the same sizes as `recv_off` arrive
  void ExchangeLocal(const double *sendbase, const std::vector<int> &send_off, double *recvbase, const std::vector<int> &recv_off,
                     hipStream_t s) const;
  std::vector<int32_t> iface_;  // every local dof that is sent or received (host copy, sorted, unique)
  // peer transport (Comm::PeerReady): device-side plan of this halo, see comm.hip
  struct PeerPlan;
  PeerPlan *peer_ = nullptr;
  void PeerSetup(const int32_t *send_idx);
  void FreePeer();
  void PeerExchange(int dir, double *d_v, hipStream_t s) const;

  std::vector<int32_t> shared_owned_;  // peer transport: the owned dofs that have sharers (sorted)
  mutable double *d_ghost_out_ = nullptr;  // direct form: the ghost rows of the local apply

public:
  bool UsesPeerTransport() const { return peer_ != nullptr; }
  // peer transport: ParOperator::Mult's two exchanges with the copies x -> lx / ly -> y and the essential-dof handling
  // folded into their kernels (see comm.hip); mask [n_true]: bit 1 essential, bit 2 owned dof with sharers
  const std::vector<int32_t> &SharedOwnedDofs() const { return shared_owned_; }
  void ProlongateFused(const double *d_x, const uint8_t *d_mask, int n_true, double *d_lx, hipStream_t s) const;
  void RestrictAddFused(const double *d_ly, const double *d_x, const uint8_t *d_mask, bool diag_one, int n_true, double *d_y,
                        hipStream_t s) const;
  // Direct form (no L-vectors; the local operator reads the ghosts from the mailbox and writes the ghost rows to GhostOut():
  // pa_op_mult_split).  DirectOk: peer transport and the ghosts are the contiguous tail [n_true, n_local) in receive order.
  bool DirectOk(int n_true, int n_local) const;
  void SendDirect(const double *d_x, const uint8_t *d_mask, hipStream_t s) const;
  const double *GhostIn(int buffer) const;             // the two mailbox buffers of P ...
  const unsigned long long *GhostInSelector() const;   // ... and the device counter whose parity names the current one
  double *GhostOut() const;
  void RestrictAddDirect(const uint8_t *d_mask, double *d_y, hipStream_t s) const;
  // The same with a smoother step consumed where the sum over the ranks is formed (round 6; linalg.hpp: Operator::MultChebyStep on
  // split vectors): for every owned dof with sharers t = t_iface[d] + the neighbours' rows (essential rows: t_iface[d] alone), then
  //   mode 1:  out[d] (+)= ek[d] + sd (ek[d] - ep[d]) + sr dinv[d] (r0[d] - t)      mode 2:  res[d] = r0[d] - t,  out[d] = sr dinv[d] (r0[d] - t)
  // -- the line the local gather evaluates for the dofs no other rank shares (pa_op_mult_split_step).  Merged form of P^T only.
  using Step = HaloStep;
  bool StepOk() const;
  void RestrictAddDirectStep(const uint8_t *d_mask, const double *d_t_iface, const Step &st, hipStream_t s) const;
  Halo(Comm &comm, int nnbr, const int *nbr, const int *send_off, const int32_t *send_idx, const int *recv_off,
       const int32_t *recv_idx);
  ~Halo();
  // the local dofs the exchange touches: elements without any of them do not depend on it
  const std::vector<int32_t> &InterfaceDofs() const { return iface_; }
  // The plan against the vector it will be used on: owned dofs sent are true dofs, ghosts lie in [n_true, n_local) -- in
  // particular the contiguous ghost range received into / sent from in place.  Called by every operator that takes a plan.
  void Validate(int n_true, int n_local) const;
  // lx[ghosts] <- owners' values   (P)
  void Prolongate(double *d_lx, hipStream_t s) const;
  // ly[owned shared] += sharers' ghost contributions   (P^T)
  void RestrictAdd(double *d_ly, hipStream_t s) const;
};

}  // namespace palace
