#!/bin/bash
# The round's evidence set on one GPU box (through gpurun): tools/r05_evidence.sh <tag>
set -u
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err      # the driver's own command line
bash tools/profile_bench.sh $TAG --no-api --no-other-configs > $O/profile_bench.log 2>&1
bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
bash tools/pmc_osd_kernel.sh $TAG > $O/pmc_osd_kernel.log 2>&1
bash tools/pmc_osd_kernel.sh ${TAG}_cs1 bb144_custom_r12_p0.003 osd_cs 1 > $O/pmc_osdcs_kernel.log 2>&1
# rocprofv3 kernel trace of the OSD-CS configuration (the reference wrapper's default post-processor)
( cd /tmp && export TMPDIR=/tmp && QD_NO_PIPELINE=1 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_osdcs -o trace -- python $GRAFT_REPO_ROOT/bench.py --osd-method osd_cs --osd-order 1 --shots 131072 --steps 3 --warmup 1 --no-cpu --no-api --no-other-configs > $GRAFT_REPO_ROOT/$O/bench_osdcs_under_trace.json 2> $GRAFT_REPO_ROOT/$O/trace_osdcs.err )
for f in $(find gpurun_out/prof_${TAG}_osdcs -name "*kernel_stats.csv"); do head -8 $f; done > $O/rocprofv3_osdcs_kernel_stats.txt
python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
QUITS_AMD_LIB=$PWD/build_ablate/lib_osdtiming.so FIXTURE=bb144_custom_r12_p0.003 python tools/osd_timing.py > $O/osd_phase_timers.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
