#!/bin/bash
# The round's evidence set on one GPU box (through gpurun): tools/r04_evidence.sh <tag>
set -u
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err      # the driver's own command line
bash tools/profile_bench.sh $TAG --no-api > $O/profile_bench.log 2>&1
bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
bash tools/pmc_osd_kernel.sh $TAG > $O/pmc_osd_kernel.log 2>&1
bash tools/pmc_osd_kernel.sh ${TAG}_p006 bb144_custom_r12_p0.006 > $O/pmc_osd_kernel_p006.log 2>&1
bash tools/r04_spot.sh $TAG > $O/spot.log 2>&1
python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so FIXTURE=bb144_custom_r12_p0.003 python tools/osd_timing.py > $O/osd_phase_timers.txt 2>&1
QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so FIXTURE=bb144_custom_r12_p0.006 python tools/osd_timing.py >> $O/osd_phase_timers.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
