#!/bin/bash
# round 6: the one-pass complex applies -- tests, timing (two-part gather on / off, affine batches on / off), counters of the
# isotropic and the anisotropic form (FETCH_SIZE / WRITE_SIZE in separate passes)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_complex
mkdir -p $OUT
python -m pytest tests/test_complex_gpu.py tests/test_parity_r02_gpu.py -x -q 2>&1 | tail -5 > $OUT/tests.log
for g in 1 0; do for a in 1 0; do
  PALACE_AMD_CPLX_GATHER2=$g PALACE_AMD_STREAM_AFFINE=$a python scripts/time_complex_r06.py > $OUT/time_gather2_${g}_affine_${a}.json 2> $OUT/time_${g}_${a}.err
done; done
cd /tmp && export TMPDIR=/tmp
for iso in 1 0; do for ctr in FETCH_SIZE WRITE_SIZE; do
  ISO=$iso rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_iso${iso}/$ctr -- python $GRAFT_REPO_ROOT/scripts/pmc_complex_aniso.py > $OUT/pmc_iso${iso}_$ctr.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/pmc_complex_aniso.py --reduce $OUT/pmc_iso${iso} > $OUT/pmc_iso${iso}.json
rm -rf $OUT/pmc_iso${iso}
done
cat $OUT/tests.log; for f in $OUT/time_*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f'))
print({k:(round(v['ms'],4), round(v['hbm_frac'],3)) for k,v in d.items() if isinstance(v,dict) and 'ms' in v})"; done; cat $OUT/pmc_iso1.json; echo; cat $OUT/pmc_iso0.json
