#!/bin/bash
# round 5, GPU call 26: where the packed complex kernel requests x of the next batch (A / B), and the real anisotropic K + M kernel's
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for g in 2 1 3 2; do
  PALACE_AMD_CPLX_GPOS=$g timeout 100 python scripts/time_complex_aniso.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CPLX_GPOS=$g', 'complex %.4f ms' % d['ms'], 'real K+M %.4f ms' % d['real_aniso_curlcurl_mass_ms'])"
done | tee gpurun_out/r05_cplx_gpos_ab.log
for g in 1 0; do
  PALACE_AMD_STREAM_GPOS=$g timeout 100 python scripts/time_complex_aniso.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('STREAM_GPOS=$g', 'complex %.4f ms' % d['ms'], 'real K+M %.4f ms' % d['real_aniso_curlcurl_mass_ms'])"
done | tee -a gpurun_out/r05_cplx_gpos_ab.log
