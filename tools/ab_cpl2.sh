# Same-box A/B of the several-checks-per-lane scatter kernel shapes through bench.py (through gpurun): tools/ab_cpl2.sh <tag>
set -u
TAG=${1:-cpl2}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "scatter or headline or smoke or config4_qlp" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
run() {   # run <switch> <bench args>
  local SW=$1; shift
  for sw in 0 1 0 1; do
    if [ $sw = 1 ]; then export $SW=1; else unset $SW; fi
    timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$SW=$sw $*', round(d['value']), round(d['ms_per_step'],2), d.get('logical_error_rate'), r['kernel'], round(r['avg_launch_ms'],2), round(r['osd_kernel_ms_per_launch'],2), round(r['frac'],3))
"
  done
  unset $SW
}
{
Q="--code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --steps 2"
run QD_SCATTER_NATURAL_ROUNDS $Q
run QD_SCATTER_WIDE_T704 $Q
run QD_SCATTER_WIDE_T384 $Q
run QD_SCATTER_CPL4
run QD_SCATTER_NATURAL_ROUNDS --code hgp225 --shots 16384 --max-iter 30
} 2>&1 | tee $O/ab.txt
