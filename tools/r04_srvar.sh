#!/bin/bash
# same-box comparison of qd_osd0_sr_kernel shapes: usage tools/r04_srvar.sh <outdir> <name> [<name> ...]
cd "$(dirname "$0")/.."
O=gpurun_out/$1; mkdir -p $O; shift
for v in "$@"; do
  for p in 0.003 0.006; do
    QUITS_AMD_LIB=$PWD/build_ablate/lib_sr_$v.so QD_NO_PIPELINE=1 timeout 300 python bench.py --p $p --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v p=$p', round(d['value']), 'bp', round(r['avg_launch_ms'],2), 'osd', round(r['osd_kernel_ms_per_launch'],2), d.get('logical_error_rate'))
" | tee -a $O/bench.txt
  done
  for f in bb144_custom_r12_p0.003 bb144_custom_r12_p0.006; do
    echo "== $v $f" >> $O/osd_timing.txt
    QUITS_AMD_LIB=$PWD/build_ablate/lib_sr_${v}_t.so FIXTURE=$f timeout 300 python tools/osd_timing.py 2>&1 | grep -v "amdgpu.ids\|^kernel info" >> $O/osd_timing.txt
  done
done
cat $O/osd_timing.txt
