# Same-box A/B: chunk size of the pipelined driver (through gpurun)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-chunk}; mkdir -p $O
for a in "" "--p 0.006" "--osd-method lsd_cs --osd-order 1"; do
  for ch in 65536 32768 98304 65536 32768 98304; do
    QD_CHUNK_SHOTS=$ch timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu --shots 393216 $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk $ch $a', round(d['value']), round(d['ms_per_step'],2), round(d['ms_per_step_with_kernel_events'],2), d.get('logical_error_rate'))
"
  done
done 2>&1 | tee $O/ab.txt
