#!/bin/bash
# round 5, GPU call 25: HBM-side traffic of the packed complex apply (two separate --pmc passes, as the guide prescribes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pmc_caniso
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 140 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_caniso/$c -- python $R/scripts/pmc_complex_aniso.py > $R/gpurun_out/pmc_caniso/$c.out 2> $R/gpurun_out/pmc_caniso/$c.err
  tail -1 $R/gpurun_out/pmc_caniso/$c.out
done
cd $R
python scripts/pmc_complex_aniso.py --reduce gpurun_out/pmc_caniso > gpurun_out/r05_complex_aniso_pmc.json; cat gpurun_out/r05_complex_aniso_pmc.json
rm -rf gpurun_out/pmc_caniso
