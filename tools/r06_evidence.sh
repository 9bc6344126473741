#!/bin/bash
# The round's evidence set on one GPU box (through gpurun): tools/r06_evidence.sh <tag>.  Every command under its own timeout.
set -u
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err      # the driver's own command line
timeout 900 bash tools/profile_bench.sh $TAG --shots 262144 --no-api --no-other-configs > $O/profile_bench.log 2>&1
timeout 600 bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
timeout 600 bash tools/pmc_osd_kernel.sh $TAG > $O/pmc_osd_kernel.log 2>&1
timeout 600 bash tools/pmc_osd_kernel.sh ${TAG}_cs1 bb144_custom_r12_p0.003 osd_cs 1 > $O/pmc_osdcs_kernel.log 2>&1
# rocprofv3 kernel trace of the OSD-CS configuration (the reference wrapper's default post-processor)
( cd /tmp && export TMPDIR=/tmp && QD_NO_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_osdcs -o trace -- python $GRAFT_REPO_ROOT/bench.py --osd-method osd_cs --osd-order 1 --shots 131072 --steps 3 --warmup 1 --no-cpu --no-api --no-other-configs > $GRAFT_REPO_ROOT/$O/bench_osdcs_under_trace.json 2> $GRAFT_REPO_ROOT/$O/trace_osdcs.err )
for f in $(find gpurun_out/prof_${TAG}_osdcs -name "*kernel_stats.csv"); do head -8 $f; done > $O/rocprofv3_osdcs_kernel_stats.txt
timeout 900 python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
for m in 1 2 3; do
  [ -f build_ablate/lib_cstiming$m.so ] && QD_CS_SUB=$m QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming$m.so timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/osdcs_phase_headline_sub$m.txt
done
[ -f build_ablate/lib_cstiming1.so ] && QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so FIXTURE=qlp1020_cardinal_r20_p0.003 WINDOW=3,1,5 SHOTS=2048 timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/osdcs_phase_qlp_sub1.txt
[ -f build_ablate/lib_cstiming1.so ] && QD_CS_SUB=1 QUITS_AMD_LIB=$PWD/build_ablate/lib_cstiming1.so WINDOW=5,3,1 timeout 300 python tools/osdcs_timing.py 2>&1 | grep -v amdgpu.ids > $O/osdcs_phase_w5f3_sub1.txt
timeout 900 python tools/stress_parity.py 2000 2026 > $O/stress_parity.txt 2>&1
tail -2 $O/stress_parity.txt
timeout 300 python tools/stress_windows.py > $O/stress_windows.txt 2>&1
tail -2 $O/stress_windows.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
