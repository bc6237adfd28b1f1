#!/bin/bash
# A/B: does the HIP event per timed step change `value`?  Alternating fresh processes, the driver's flags, no extras.
for i in 1 2 3; do
  for ev in 1 0; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --after-idle-launches 0 --step-events $ev | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('step_events', $ev, 'value %.3f T' % (d['value'] / 1e12), 'ms_per_step %.4f' % d['ms_per_step'], 'steady %.3f T' % (d['config']['steady_pairs_per_s'] / 1e12), 'kernel_ms %.4f' % d['roofline']['kernel_ms_avg'])"
  done
done
