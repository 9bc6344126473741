#!/usr/bin/env python3
"""Counters of the one-wavefront-per-shot OSD-0 kernel (csrc/osd_wave.hip) on the headline window; needs the three
-DQD_OSD_TIMING -DQW_DBG_MODE={1,2,3} builds (QUITS_AMD_LIB=build_ablate/lib_wavedbg<mode>.so MODE=<mode>).  GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
name = os.environ.get("FIXTURE", "bb144_custom_r12_p0.003")
H, L, pri = helpers.dem_matrices(name)
det, obs = DemSampler(H, L, pri).sample(32768, seed=5)
g = WindowGraph(H, pri); d = BatchDecoder(g, max_iter=50, osd_method="osd_0")
d.decode(det, stage=1); torch.cuda.synchronize(); d.debug_counters()
d.decode(det, stage=2); torch.cuda.synchronize()
c = d.debug_counters()[12:16]
mode = int(os.environ.get("MODE", "1"))
if mode == 1:
    print("shots %d  handed over %d (%.1f %%)  columns per shot %.1f  gave up early (ties / 128 pivots) %d" % (c[0], c[1], 100.0 * c[1] / max(c[0], 1), c[2] / max(c[0], 1), c[3]))
elif mode == 2:
    n = max(c[3], 1)
    print("ticks per shot (10 ns): threshold %.0f   gather + sort + column fetch %.0f   elimination %.0f" % (c[0] / n, c[1] / n, c[2] / n))
else:
    n = max(c[3], 1)
    print("prefix %.1f columns  pivots %.1f  columns that needed the row scan %.1f   per shot" % (c[0] / n, c[1] / n, c[2] / n))
