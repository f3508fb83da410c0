#!/bin/bash
# round 5, GPU calls 30 / 32: what the ranks of the two-process BOOMER_AMG / AMS tests hold (PALACE_AMD_COARSE_VERBOSE=1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/sr
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O2 -w -I$R/palace_amd/csrc -I$R/include $R/examples/cxx_host/solve_ranks.cpp -L$R/palace_amd/lib -lpalace_amd -Wl,-rpath,$R/palace_amd/lib -o /tmp/sr/solve_ranks
export HSA_ENABLE_IPC_MODE_LEGACY=0 PALACE_AMD_PEER_TIMEOUT_S=30 PALACE_AMD_COARSE_VERBOSE=1
{
python examples/cxx_host/dump_problem_ranks.py /tmp/sr/amg 2 2 10 20; mkdir -p /tmp/sr/h1
for r in 0 1; do timeout 100 /tmp/sr/solve_ranks /tmp/sr/amg $r 2 /tmp/sr/h1 amg & done; wait
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dist_levels.log
