#!/usr/bin/env python3
"""How much of the scatter kernel's work repeats itself: share of (wavefront, round, iteration) triples in which no check of the
wavefront changes what it sends (needs a -DQSW_STATS build: QUITS_AMD_LIB=build_ablate/lib_bpstats.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
name = os.environ.get("FIXTURE", "bb144_custom_r12_p0.003")
H, L, pri = helpers.dem_matrices(name)
det, obs = DemSampler(H, L, pri).sample(16384, seed=5)
g = WindowGraph(H, pri); d = BatchDecoder(g, max_iter=50, osd_method="osd_0")
d.decode(det, stage=1); torch.cuda.synchronize(); d.debug_counters()
d.decode(det, stage=1); torch.cuda.synchronize()
c = d.debug_counters()
print(name, "wave-round-iterations", c[0], "with no change", c[1], "= %.3f" % (c[1] / max(c[0], 1)))
for b in range(5):
    print("  iterations %d..%d: %d, unchanged share %.3f" % (10 * b, 10 * b + 9, c[2 + b], c[7 + b] / max(c[2 + b], 1)))
print("  checks unchanged share %.3f" % (c[12] / max(c[13], 1)))
