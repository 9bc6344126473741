# Parity evidence of the final binary at scale (through gpurun): randomised graph / option sweep against the oracle, all window shapes,
# and the 2 x 10^6 shots of the round-2 LER study decoded again (failure bits must equal the oracle's double-precision grid column)
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
O=gpurun_out/${1:-parity}; mkdir -p $O
for seed in 41 42 43; do timeout 600 python tools/stress_parity.py 700 $seed 2>&1 | tail -1; done | tee $O/stress_parity.txt
timeout 600 python tools/stress_windows.py 2>&1 | tail -1 | tee -a $O/stress_parity.txt
timeout 900 python tools/recheck_ler_forms.py 2>&1 | tail -6 | tee $O/recheck_ler_forms.txt
