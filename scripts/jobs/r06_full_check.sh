#!/bin/bash
# the whole GPU suite, smoke(), and the bench line on the current library
cd $GRAFT_REPO_ROOT
(time python -m pytest tests/ -x -q -m gpu) > gpurun_out/r06_gpu_tests_final.log 2>&1
tail -3 gpurun_out/r06_gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1 | tee -a gpurun_out/r06_gpu_tests_final.log
python bench.py > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err
tail -c 600 gpurun_out/r06_bench_n1.json
