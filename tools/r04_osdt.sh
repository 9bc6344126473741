#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04t}; mkdir -p $O
for f in bb144_custom_r12_p0.003 bb144_custom_r12_p0.006; do
  echo "== $f osd_0 $2" >> $O/osd_timing.txt
  env $2 QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so FIXTURE=$f timeout 300 python tools/osd_timing.py 2>&1 | grep -v amdgpu.ids >> $O/osd_timing.txt
done
cat $O/osd_timing.txt
