# Lean evidence pass for the current binary on one GPU box (through gpurun): tools/head_evidence.sh <tag>
set -u
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err      # the driver's own command line
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1
bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
for a in "--code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --steps 2 --warmup 1" "--code bb72 --window 3 1 --steps 3 --warmup 1" "--window 5 3 --steps 3 --warmup 1" "--window 3 1 --steps 3 --warmup 1"; do
  python bench.py --no-cpu $a 2>/dev/null | tail -1
done > $O/spot_configs.jsonl
QD_SCATTER_SMALL=1 python bench.py --no-cpu --window 3 1 --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/spot_small.jsonl
timeout 900 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
