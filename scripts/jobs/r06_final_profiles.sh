#!/bin/bash
# round 6, final library: the rocprofv3 evidence of the round (profile_round.sh), the per-kernel-class tables of one PCG
# iteration (plain and Hiptmair smoothers), the per-rank proxy of the 8-GPU case
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh > gpurun_out/r06_profile_round.log 2>&1
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/scripts/pcg_kernel_classes.py $GRAFT_REPO_ROOT/gpurun_out/r06_pcg_classes_cheb 0 > $GRAFT_REPO_ROOT/gpurun_out/r06_pcg_classes_cheb.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pcg_kernel_classes.py $GRAFT_REPO_ROOT/gpurun_out/r06_pcg_classes_hiptmair 1 > $GRAFT_REPO_ROOT/gpurun_out/r06_pcg_classes_hiptmair.log 2>&1
cd $GRAFT_REPO_ROOT
(PCG=50 python scripts/time_halo_mult.py 2>&1 | sed 's/^/[peer] /') > gpurun_out/r06_halo_proxy.log
(HALO_MODE=none PCG=50 python scripts/time_halo_mult.py 2>&1 | sed 's/^/[no halo] /') >> gpurun_out/r06_halo_proxy.log
head -3 gpurun_out/r06_pcg_classes_cheb.log; head -3 gpurun_out/r06_pcg_classes_hiptmair.log; cat gpurun_out/r06_halo_proxy.log | cut -c1-260; tail -3 gpurun_out/r06_profile_round.log
