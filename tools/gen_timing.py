#!/usr/bin/env python3
"""Phase breakdown of the serial schedule in the general BP kernel (needs a -DQD_GEN_TIMING build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
det, obs = DemSampler(H, L, pri).sample(N, seed=5)
g = WindowGraph(H, pri)
d = BatchDecoder(g, bp_method=sys.argv[2] if len(sys.argv) > 2 else "product_sum", schedule="serial", max_iter=10, osd_method="osd_0")
d.decode(det, stage=1); torch.cuda.synchronize(); d.debug_counters()
d.set_profiling(True); d.decode(det, stage=1); torch.cuda.synchronize()
c = d.debug_counters(); pr = d.profile()
tot = sum(c[:4]) or 1
print("bp ms", pr["bp_ms"], "lane-0 iterations", c[4])
for i, nme in enumerate(["adjacency (scalar loads)", "row scans", "log + posterior", "backward sweep + loop"]):
    print("%-26s %5.1f %%  %10.0f ticks per lane-0 iteration" % (nme, 100.0 * c[i] / tot, c[i] / max(c[4], 1)))
