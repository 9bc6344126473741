#!/bin/bash
# timelines of two more workloads: usage tools/r06_tl2.sh <outdir>
cd "$(dirname "$0")/.."
bash tools/r06_stage_trace.sh $1/w3f1 --window 3 1 --shots 262144 > /dev/null
python tools/r06_timeline.py gpurun_out/$1/w3f1/trace 3000 | grep -v "at::native" > gpurun_out/$1/w3f1/timeline.txt
bash tools/r06_stage_trace.sh $1/oscs --osd-method osd_cs --osd-order 1 --shots 262144 > /dev/null
python tools/r06_timeline.py gpurun_out/$1/oscs/trace 600 | grep -v "at::native" > gpurun_out/$1/oscs/timeline.txt
rm -rf gpurun_out/$1/*/trace
