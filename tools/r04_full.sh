#!/bin/bash
# full GPU suite + randomised parity sweeps + bench lines; usage tools/r04_full.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04full}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gputests.txt; cat $O/gputests.txt
timeout 900 python tools/stress_parity.py ${2:-600} 404 2>&1 | tail -6 > $O/stress_parity.txt; cat $O/stress_parity.txt
timeout 600 python tools/stress_windows.py 2>&1 | tail -4 > $O/stress_windows.txt; cat $O/stress_windows.txt
timeout 300 python bench.py --no-cpu > $O/bench_nocpu.json 2>/dev/null; tail -c 1500 $O/bench_nocpu.json
