#!/bin/bash
# round 5, GPU call 7: dense affine kernel A/B -- waves per workgroup (12 | 8) x prefetch distance of x (one block | before the products)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" _w12x0 _w8x1 _w8x0; do
  export PALACE_AMD_LIB=$PWD/palace_amd/lib/libpalace_amd$v.so
  echo "== lib${v:-_default(w12x1)}"
  ( N=36 timeout 300 python scripts/time_tet.py ) 2>&1 | grep "mult" | cut -c1-120
done > gpurun_out/r6_tet_ab.log 2>&1
cat gpurun_out/r6_tet_ab.log
for v in "" _w8x1; do
  export PALACE_AMD_LIB=$PWD/palace_amd/lib/libpalace_amd$v.so
  ( timeout 600 python -m pytest -q -x -m gpu tests/test_tet_gpu.py tests/test_split_gpu.py ) 2>&1 | tail -2
done
