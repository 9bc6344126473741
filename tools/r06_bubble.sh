#!/bin/bash
# the BP stream's per-chunk bubble: counter fills at the head of the BP stage (QD_COUNTER_FILLS=1, the round-5 order) against counters zeroed by the
# stage before on its own stream; usage tools/r06_bubble.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for rep in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export QD_COUNTER_FILLS=1; else unset QD_COUNTER_FILLS; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-api --no-other-configs 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fills at the head of the BP stage: $v', 'shots/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2))
" | tee -a $O/bench.txt
done
done
unset QD_COUNTER_FILLS
bash tools/r06_stage_trace.sh $1/trace_new --shots 1048576 > /dev/null
python tools/r06_timeline.py gpurun_out/$1/trace_new/trace 400 | grep -v "at::native" > $O/timeline_new.txt
