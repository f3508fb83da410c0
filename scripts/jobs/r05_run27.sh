#!/bin/bash
# round 5, GPU call 27: pa_replicated_coarse_info through the two-process test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_peer_gpu.py -q -m gpu -k "assembled_from_the_ranks_pieces" 2>&1 | tail -3 | tee gpurun_out/r05_run27.log
