#!/bin/bash
# round 5, GPU call 15: geometry from the nodes in the streaming curl-curl kernel: result against the packed form, timing of four variants
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM" ) > gpurun_out/r13_geomn.log
for v in w3g2 w3g1 w2g2 w2g1; do
  ( PALACE_AMD_STREAM_GEOM=nodes PALACE_AMD_GEOMN_VARIANT=$v timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM\|rror" | tail -3 ) >> gpurun_out/r13_geomn.log
done
( timeout 300 python scripts/geomn_ab.py 2>&1 | grep "GEOM" ) >> gpurun_out/r13_geomn.log
cat gpurun_out/r13_geomn.log | cut -c1-250
