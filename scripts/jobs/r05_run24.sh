#!/bin/bash
# round 5, GPU call 24: the whole GPU suite on the final library (after the transport set-up changes), as the driver runs it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 700 python -m pytest tests -x -q -m gpu ) > gpurun_out/r05_gpu_tests_final2.log 2>&1
grep -n "passed\|failed" gpurun_out/r05_gpu_tests_final2.log | tail -2
python -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
