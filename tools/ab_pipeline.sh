# Same-box A/B of the pipelined driver (a chunk's post-processing beside the other chunk's BP) against QD_NO_PIPELINE=1 (through gpurun)
set -u
TAG=${1:-pipe}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "pipelined or ler_agreement or smoke or published or sliding" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
run() {
  local SW=$1; shift
  for sw in 1 0 1 0; do
    if [ $sw = 1 ]; then export $SW=1; else unset $SW; fi
    timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$SW=$sw $*', round(d['value']), round(d['ms_per_step'],2), round(d['ms_per_step_with_kernel_events'],2), d.get('logical_error_rate'), r['kernel'], round(r['avg_launch_ms'],2), round(r['osd_kernel_ms_per_launch'],2), d['pipeline']['osd_beside_next_chunk_bp'])
"
  done
  unset $SW
}
{
run QD_NO_PIPELINE
run QD_NO_PIPELINE --window 5 3
run QD_NO_PIPELINE --window 3 1
run QD_NO_PIPELINE --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --window 5 3 --shots 327680 --steps 2
run QD_NO_PIPELINE --bp-method product_sum --schedule serial --max-iter 10 --osd-method lsd_cs --osd-order 1 --window 5 3 --shots 327680 --steps 2
run QD_NO_PIPELINE --code qlp1020 --window 3 1 --shots 131072 --p-override 0.001 --steps 2
} 2>&1 | tee $O/ab.txt
