#!/bin/bash
# A/B of an ENVIRONMENT variable read by the library at first use (process-wide), on the C5 RANSAC batch: alternating fresh processes.
#   bash tools/c5_env_ab.sh MLPL_TRI_WPE 4 5 6
VAR=$1; shift
for i in 1 2 3; do
  for v in "$@"; do
    env $VAR=$v python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 3 | grep '^bench_detail ' | python -c "
import sys, json
d = json.loads(sys.stdin.read()[len('bench_detail '):])
print('$VAR', $v, 'ms_per_step %.3f' % d['ms_per_step'], d['mode'], {k[:24]: round(x, 3) for k, x in d['kernel_ms_per_step_rank0'].items()})"
  done
done
