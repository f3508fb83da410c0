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
  void Arrive();  // barrier over the ranks (threads)
};

class Comm {
  int rank_ = 0, size_ = 1;
  void *nccl_ = nullptr;  // ncclComm_t
  LocalGroup *local_ = nullptr;
  friend class Halo;

public:
  static constexpr int kUniqueIdBytes = 128;
  static void GetUniqueId(char *out);
  Comm(int rank, int size, const char *unique_id);
  Comm(int rank, LocalGroup &group);  // rank of an in-process group (see LocalGroup)
  ~Comm();
  int Rank() const { return rank_; }
  int Size() const { return size_; }
  // in-place sum over ranks of n doubles in device memory (Mpi::GlobalSum)
  void AllReduceSum(double *d_buf, int n, hipStream_t s);
  void Barrier(hipStream_t s);
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
  // in-process group: pieces [off[k], off[k + 1]) of `sendbase` go to neighbour k, pieces of the same sizes as `recv_off` arrive
  void ExchangeLocal(const double *sendbase, const std::vector<int> &send_off, double *recvbase, const std::vector<int> &recv_off,
                     hipStream_t s) const;
  std::vector<int32_t> iface_;  // every local dof that is sent or received (host copy, sorted, unique)

public:
  Halo(Comm &comm, int nnbr, const int *nbr, const int *send_off, const int32_t *send_idx, const int *recv_off,
       const int32_t *recv_idx);
  ~Halo();
  // the local dofs the exchange touches: elements without any of them do not depend on it
  const std::vector<int32_t> &InterfaceDofs() const { return iface_; }
  // lx[ghosts] <- owners' values   (P)
  void Prolongate(double *d_lx, hipStream_t s) const;
  // ly[owned shared] += sharers' ghost contributions   (P^T)
  void RestrictAdd(double *d_ly, hipStream_t s) const;
};

}  // namespace palace
