#!/bin/bash
# K1g: the serial schedule in several launches with the survivors packed in between (GenStage), against one launch (QD_GEN_STAGES=0), same box
# usage tools/r06_stage_ab.sh <outdir>
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "general or reference_default or product_sum or bplsd or phenom" > $O/tests.txt 2>&1; tail -5 $O/tests.txt
B="--steps 3 --warmup 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2; do
for v in "" 0 "3,5,7" "3,6,8"; do
  QD_GEN_STAGES_SET=$v
  if [ -z "$v" ]; then unset QD_GEN_STAGES; else export QD_GEN_STAGES=$v; fi
  timeout 300 python bench.py --window 5 3 --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --shots 163840 $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('stages [${v:-default}]', 'shots/s', round(d['value']), 'LER', d['logical_error_rate'], 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2))
" | tee -a $O/bench.txt
done
done
unset QD_GEN_STAGES
for v in "" 0; do
  if [ -z "$v" ]; then unset QD_GEN_STAGES; else export QD_GEN_STAGES=$v; fi
  timeout 300 python bench.py --code hgp225 --window 3 1 --bp-method product_sum --schedule serial --osd-method osd_cs --osd-order 1 --max-iter 10 --shots 65536 $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('hgp225 W=3 F=1 reference settings, stages [${v:-default}]', 'shots/s', round(d['value']), 'LER', d['logical_error_rate'], 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2))
" | tee -a $O/bench.txt
  timeout 300 python bench.py --osd-method lsd_cs --osd-order 1 --bp-method product_sum --schedule serial --max-iter 30 --window 3 1 --shots 163840 $B 2>>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bb144 W=3 F=1 product_sum serial max_iter 30 lsd_cs(1), stages [${v:-default}]', 'shots/s', round(d['value']), 'LER', d['logical_error_rate'], 'BP ms', round(r.get('avg_launch_ms') or 0, 2), 'post ms', round(r.get('osd_kernel_ms_per_launch') or 0, 2))
" | tee -a $O/bench.txt
done
