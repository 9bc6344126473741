set -u
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
mkdir -p gpurun_out/r02e
python bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
tail -c 600 gpurun_out/r02e/bench.json
bash tools/profile_bench.sh r02e > gpurun_out/r02e/profile_bench.log 2>&1
bash tools/run_configs.sh > gpurun_out/r02e/other_configs.jsonl 2> gpurun_out/r02e/other_configs.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/trace_lsd -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --osd-method lsd_0 > $GRAFT_REPO_ROOT/gpurun_out/r02e/bench_lsd_trace.json 2>/dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/trace_osdcs -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --osd-method osd_cs --osd-order 1 --shots 32768 > $GRAFT_REPO_ROOT/gpurun_out/r02e/bench_osdcs_trace.json 2>/dev/null)
for f in $(find gpurun_out/r02e/trace_lsd gpurun_out/r02e/trace_osdcs -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done > gpurun_out/r02e/trace_other_summary.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r02e/gputests.txt
cat gpurun_out/r02e/gputests.txt
