# Lean evidence pass for the current binary on one GPU box (through gpurun): tools/head_evidence.sh <tag>
set -u
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1
bash tools/pmc_bp_kernel.sh $TAG > $O/pmc_bp_kernel.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > $O/gputests.txt 2>&1
tail -3 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
