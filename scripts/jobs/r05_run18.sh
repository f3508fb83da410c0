#!/bin/bash
# round 5, GPU call 18: the packed-D complex kernel (anisotropic hexahedra), the streaming form of the anisotropic K + M operator,
# surface terms after the fused pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_complex_gpu.py tests/test_apply_gpu.py tests/test_sum_gpu.py tests/test_stream5_gpu.py tests/test_cpw_gpu.py tests/test_dense_gpu.py tests/test_cxx_boundary_gpu.py -x -q -m gpu > gpurun_out/r05_run18_tests.log 2>&1
tail -5 gpurun_out/r05_run18_tests.log
PARITY=1 timeout 300 python scripts/time_complex_aniso.py > gpurun_out/r05_complex_aniso_fused.json 2> gpurun_out/r05_complex_aniso_fused.err
PALACE_AMD_COMPLEX_FUSED=0 timeout 300 python scripts/time_complex_aniso.py > gpurun_out/r05_complex_aniso_unfused.json 2> gpurun_out/r05_complex_aniso_unfused.err
cat gpurun_out/r05_complex_aniso_fused.json gpurun_out/r05_complex_aniso_unfused.json; tail -3 gpurun_out/r05_complex_aniso_fused.err
