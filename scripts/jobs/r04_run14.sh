#!/bin/bash
# round 4, GPU call 14: the cpw leg with the batched (CGS2) orthogonalisation next to the reference's MGS
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/profile_cpw.py 2>&1 | tail -2 | cut -c1-1500 | tee gpurun_out/r14_cpw.log
