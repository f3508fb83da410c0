#!/bin/bash
# round 5, GPU call 5: stage ablation of the resident dense (tetrahedral) kernel + its kernel table
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( PALACE_AMD_LIB=$PWD/palace_amd/lib/libpalace_amd_ablate.so timeout 600 python scripts/ablate_tet.py ) > gpurun_out/r4_ablate.log 2>&1
cat gpurun_out/r4_ablate.log | cut -c1-200
