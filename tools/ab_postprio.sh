# Same-box A/B: priority of the post-processing stream of the pipelined driver (through gpurun)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-postprio}; mkdir -p $O
for a in "" "--p 0.006" "--osd-method lsd_cs --osd-order 1" "--window 3 1"; do
  for pr in 0 -1 0 -1; do
    QD_POST_STREAM_PRIORITY=$pr timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('post stream priority $pr $a', round(d['value']), round(d['ms_per_step'],2), round(d['ms_per_step_with_kernel_events'],2), d.get('logical_error_rate'))
"
  done
done 2>&1 | tee $O/ab.txt
