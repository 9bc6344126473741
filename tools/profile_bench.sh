#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes for HBM traffic of bench.py.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are those of kernels that have the GPU to themselves (as bench.py's roofline pass): no overlap of
# the post-processing with the next chunk's BP in these runs
export QD_NO_PIPELINE=1
ARGS="--steps 3 --warmup 1 --no-cpu $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o write -- python $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
python $REPO/tools/summarize_pmc.py $OUT || true
