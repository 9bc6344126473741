#!/bin/bash
# last refresh of the round's bench lines and test log on the final tree (kernels unchanged since tools/r04_evidence.sh): tools/r04_final.sh <tag>
set -u
TAG=${1:-r04f}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
