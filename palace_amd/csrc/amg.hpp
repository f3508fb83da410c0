// Host-side set-up of a smoothed-aggregation hierarchy for an assembled (CSR) coarse-level matrix.
//
// The reference hands its coarsest p-multigrid level to HYPRE (BoomerAMG for H1 problems, AMS for H(curl) ones:
// linalg/amg.cpp:12-49, linalg/ams.cpp:18-224) -- third-party code that is not part of /root/reference.  This is the
// set-up half of the native replacement SURVEY.md 8 f3 asks for: strength graph, greedy aggregation, tentative and
// Jacobi-smoothed prolongators, Galerkin products.  It runs once per operator on the host, on the matrix
// pa_op_full_assemble produces (p = 1 levels: a few 10^5 rows); the solve half runs on the device (amg_solver.hpp:
// AmgSolver = V-cycle over this hierarchy, AmsSolver = the two-space preconditioner for H(curl) built on it).
// Checked on the CPU by tests/test_fem_host.py (tests/cpu/fem_host_check.cpp) and on the GPU by tests/test_ams_gpu.py.
#pragma once

#include <vector>

namespace palace::amg {

struct HostCsr {
  int nrows = 0, ncols = 0;
  std::vector<int> rowptr, col;  // columns sorted within a row
  std::vector<double> val;
  long long nnz() const { return (long long)col.size(); }
};

HostCsr Transpose(const HostCsr &A);
HostCsr Multiply(const HostCsr &A, const HostCsr &B);  // C = A B, exact zeros kept out
void Mult(const HostCsr &A, const std::vector<double> &x, std::vector<double> &y);

// Standard aggregation on the strength graph |a_ij| >= theta sqrt(a_ii a_jj): pass 1 forms an aggregate from every node
// whose strong neighbourhood is still free, pass 2 attaches the remaining nodes to the neighbouring aggregate they are
// most strongly tied to (isolated nodes become their own aggregates).  Returns the aggregate of each row.
std::vector<int> Aggregate(const HostCsr &A, double theta, int &num_aggregates);
// The same confined to blocks of consecutive rows (the rows a rank owns, block_off [nblocks + 1]): ties across blocks are not
// strength ties, so that every aggregate lies inside one block, and the aggregates are numbered block by block
// (agg_off [nblocks + 1]: the coarse rows of a block are consecutive again -- the next level's blocks).
std::vector<int> AggregateBlocks(const HostCsr &A, double theta, const std::vector<int> &block_off, int &num_aggregates,
                                 std::vector<int> &agg_off);

// Piecewise-constant prolongator with normalised columns: T^T T = I
HostCsr TentativeProlongator(const std::vector<int> &aggregate, int num_aggregates);

// P = (I - omega D^-1 A_F) T with the filtered matrix A_F (weak off-diagonal entries lumped onto the diagonal).
// omega <= 0: omega = 4 / (3 rho) with rho an estimate of the spectral radius of D^-1 A_F (a few power iterations; the
// classical 2/3 is this value for rho = 2, which Poisson-like matrices have and Galerkin products of H(curl) matrices do not)
HostCsr SmoothProlongator(const HostCsr &A, const HostCsr &T, double theta, double omega);
// rows of the flagged dofs emptied: essential dofs taken out of a transfer matrix
HostCsr DropRows(const HostCsr &A, const std::vector<char> &flag);

struct Hierarchy {
  std::vector<HostCsr> A;  // A[0] the input, A[l+1] = P[l]^T A[l] P[l]
  std::vector<HostCsr> P;  // P[l]: level l+1 -> level l
};
// Levels are added until a level has at most `coarse_size` rows, stops coarsening, or `max_levels` is reached
Hierarchy Setup(const HostCsr &A, int max_levels = 10, int coarse_size = 200, double theta = 0.08, double omega = 0.0);
// Hierarchy for a row-distributed solve: aggregation confined to the ranks' row blocks (AggregateBlocks), prolongator smoothing and
// Galerkin products on the whole matrix as above.  level_off[l] [nblocks + 1]: the row blocks of level l.
Hierarchy SetupBlocks(const HostCsr &A, const std::vector<int> &block_off, std::vector<std::vector<int>> &level_off, int max_levels = 10,
                      int coarse_size = 200, double theta = 0.08, double omega = 0.0);

}  // namespace palace::amg
