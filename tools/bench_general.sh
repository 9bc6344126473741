#!/bin/bash
# bench.py lines for the general (one message per edge) BP kernel at the headline code.  usage: tools/bench_general.sh [shots]
SHOTS=${1:-65536}
for cfg in "product_sum serial 10" "product_sum parallel 50" "minimum_sum serial 10"; do set -- $cfg
  timeout 600 python bench.py --steps 1 --warmup 1 --shots $SHOTS --bp-method $1 --schedule $2 --max-iter $3 --cpu-shots ${CPU_SHOTS:-0} ${NOCPU:---no-cpu} 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', 'shots/s %.0f' % d['value'], 'LER %.4f' % d['logical_error_rate'], 'conv %.3f iters %.2f' % (d['bp_converged_frac'], d['mean_bp_iters']), 'bp_ms %.1f osd_ms %.1f' % (r['avg_launch_ms'], r['osd_kernel_ms_per_launch']), 'algGB/s %.0f' % r['achieved'])"
done
