# The round's closing measurements on one GPU box (through gpurun): tools/final_run.sh <tag>
set -u
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1
bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
bash tools/run_configs.sh > $O/other_configs.jsonl 2> $O/other_configs.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace_ref -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --window 5 3 > $GRAFT_REPO_ROOT/$O/bench_ref_settings_trace.json 2>/dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace_lsd -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --osd-method lsd_cs --osd-order 1 > $GRAFT_REPO_ROOT/$O/bench_lsd_trace.json 2>/dev/null)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace_osdcs -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --osd-method osd_cs --osd-order 1 --shots 32768 > $GRAFT_REPO_ROOT/$O/bench_osdcs_trace.json 2>/dev/null)
for f in $(find $O/trace_ref $O/trace_lsd $O/trace_osdcs -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done > $O/trace_other_summary.txt
python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
