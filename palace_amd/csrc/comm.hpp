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

namespace palace {

class Comm {
  int rank_ = 0, size_ = 1;
  void *nccl_ = nullptr;  // ncclComm_t
  friend class Halo;

public:
  static constexpr int kUniqueIdBytes = 128;
  static void GetUniqueId(char *out);
  Comm(int rank, int size, const char *unique_id);
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
