#!/bin/bash
# round 5, GPU call 19: geometry from the nodes with the partial sums parked in LDS (three waves per SIMD)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM" ) > gpurun_out/r17_geomn.log
for v in park3g2 park3g1 w2g1; do
  ( PALACE_AMD_STREAM_GEOM=nodes PALACE_AMD_GEOMN_VARIANT=$v timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM\|rror" | tail -3 ) >> gpurun_out/r17_geomn.log
done
( timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM" ) >> gpurun_out/r17_geomn.log
cat gpurun_out/r17_geomn.log | cut -c1-250
