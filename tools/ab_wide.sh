# Same-box A/B of the two-checks-per-lane scatter kernel (bp_scatter_wide.hip) against the gather kernel on the configs[4] windows
# (through gpurun): tools/ab_wide.sh <tag>
set -u
TAG=${1:-wide}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "wide_scatter or config4_qlp or scatter_kernel_and" > $O/tests.txt 2>&1
tail -5 $O/tests.txt
B="python bench.py --code qlp1020 --window 3 1 --shots 8192 --steps 2 --warmup 1 --no-cpu"
for a in "--p-override 0.001" "--p-override 0.0005" ""; do
  for sw in 0 1 0 1; do
    if [ $sw = 1 ]; then export QD_NO_SCATTER_WIDE=1; else unset QD_NO_SCATTER_WIDE; fi
    timeout 600 $B $a 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('QD_NO_SCATTER_WIDE=$sw $a', round(d['value']), round(d['ms_per_step'],1), d.get('logical_error_rate'), r['kernel'], round(r['avg_launch_ms'],2), round(r['osd_kernel_ms_per_launch'],2), round(r['frac'],3))
"
  done
done 2>&1 | tee $O/ab.txt
