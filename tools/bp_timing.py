#!/usr/bin/env python3
"""Phase breakdown of the BP kernel as wavefront 0 sees it (needs a -DQD_BP_TIMING build:
QUITS_AMD_LIB=build_ablate/lib_bptiming.so python tools/bp_timing.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
det, obs = DemSampler(H, L, pri).sample(32768, seed=5)
g = WindowGraph(H, pri); d = BatchDecoder(g, max_iter=50, osd_method="osd_0")
d.decode(det, stage=1); torch.cuda.synchronize(); d.debug_counters()
d.set_profiling(True); d.decode(det, stage=1); torch.cuda.synchronize()
c = d.debug_counters(); pr = d.profile()
names = ["check pass", "block-OR", "bit pass", "barrier", "prologue", "epilogue"]
tot = sum(c[:6]) or 1
shots, iters = max(c[6], 1), max(c[7], 1)
print("bp kernel ms", pr["bp_ms"], "shots (non-zero syndromes)", c[6], "mean iterations", iters / shots)
for i, nme in enumerate(names):
    per = c[i] / (iters if i < 4 else shots)
    print("%-12s %6.1f %%   %8.0f ticks per %s" % (nme, 100.0 * c[i] / tot, per, "iteration" if i < 4 else "shot"))
print("(ticks are clock64 = s_memtime)")
