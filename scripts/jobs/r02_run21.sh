#!/bin/bash
set -u
export PYTHONPATH=$(pwd)
for cfg in "100 10" "100 10" "1000 500" "100 10"; do
  set -- $cfg
  python bench.py --no-cpu --no-tets --no-p4 --pcg-iters 0 --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('steps', d['steps'], 'warmup', d['warmup'], 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'stream %.0f' % d['roofline']['measured_stream_GBps'])"
done
python scripts/time_apply.py 2>&1 | grep -v amdgpu.ids | tail -5
