#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HALO_MODE=peer REPS=200 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_peer -o peer -- python scripts/time_halo_mult.py > gpurun_out/r03_prof_peer.log 2>&1
tail -3 gpurun_out/r03_prof_peer.log
find gpurun_out/prof_peer -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_peer -name "*kernel_stats.csv" | head -1)
head -20 "$f" | cut -c1-200
timeout 600 python -m pytest tests/test_peer_gpu.py -x -q -m gpu 2>&1 | tail -2
HALO_MODE=peer PCG=50 timeout 600 python scripts/time_halo_mult.py 2>&1 | grep -v amdgpu.ids
