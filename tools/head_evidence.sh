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
for a in "--osd-method lsd_cs --osd-order 1" "--osd-method osd_cs --osd-order 1 --steps 2" "--window 5 3" "--window 3 1" "--code bb72" "--code bb72 --window 3 1" "--code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --steps 2"; do
  python bench.py --no-cpu --steps 3 --warmup 1 $a 2>/dev/null | tail -1
done > $O/spot_configs.jsonl
python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
timeout 1200 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
