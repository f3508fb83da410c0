#!/bin/bash
# round 5, GPU call 11: price of on-chip assembly inside a wave's 4 / a workgroup's 8 elements (wrong results, right bytes)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 0 4 8 0; do PALACE_AMD_PRICE_BLOCK=$g timeout 300 python scripts/price_block.py 2>&1 | grep "PRICE" ; done > gpurun_out/r9_price.log 2>&1
cat gpurun_out/r9_price.log | cut -c1-250
