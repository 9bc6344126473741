# Sustained run + every other config on the final binary (through gpurun): tools/sustained.sh <tag>
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/${1:-sus}; mkdir -p $O
python bench.py --steps 80 --warmup 3 --shots 131072 --no-cpu > $O/sustained_1e7.json 2> $O/sustained.err
tail -c 200 $O/sustained_1e7.json; echo
bash tools/run_configs.sh > $O/other_configs.jsonl 2> $O/other_configs.err
python tools/p_sweep.py --shots 1048576 > $O/p_sweep_1e6.jsonl 2> $O/p_sweep.err
wc -l $O/other_configs.jsonl $O/p_sweep_1e6.jsonl
