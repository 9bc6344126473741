#!/usr/bin/env python3
"""Phase breakdown of the fast OSD kernel (needs a -DQD_OSD_TIMING build: QUITS_AMD_LIB=build_ablate/lib_osdtiming.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
name = os.environ.get("FIXTURE", "bb144_custom_r12_p0.003")
if os.environ.get("WINDOW"):                      # WINDOW=W,F,k: window k of the reference's spacetime(W, F) for the fixture
    W_, F_, k_ = (int(v) for v in os.environ["WINDOW"].split(","))
    win = helpers.window_set(name, W_, F_)[k_]
    H, pri = win["H"], win["priors"]
    L = H[:8]                                     # (the window's commit matrix covers fewer columns; the observables are not used here)
elif os.path.exists(os.path.join(helpers.GOLD, "windows", name + ".npz")):
    H, L, pri = helpers.dem_matrices(name)
else:
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import Circuit
    H, L, pri = detector_error_model_to_matrix(Circuit(helpers.circuit_text(name)))
det, obs = DemSampler(H, L, pri).sample(int(os.environ.get("SHOTS", "32768")), seed=5)
g = WindowGraph(H, pri); d = BatchDecoder(g, max_iter=50, osd_method=os.environ.get("OSD_METHOD", "osd_0"), osd_order=int(os.environ.get("OSD_ORDER", "0")))
d.decode(det); torch.cuda.synchronize(); d.debug_counters()
d.set_profiling(True); d.decode(det); torch.cuda.synchronize()
c = d.debug_counters(); pr = d.profile()
names = ["tier(select+sort)", "batch-load", "pivots(rest)", "finish", "round: key+reduce", "round: barrier", "round: read partials+pivot row", "round: update"]
tot = (sum(c[:8]) + c[10]) or 1
print("%-10s %6.1f %%   %8.0f ticks/shot" % ("shot init", 100.0 * c[10] / tot, c[10] / max(c[8], 1)))

print("per shot: rounds %.2f tiers %.2f batches %.2f" % (c[11] / max(c[8], 1), c[12] / max(c[8], 1), c[13] / max(c[8], 1)));print("column kernel: batches per shot before / after npiv >= m - 64: %.1f / %.1f; ticks per shot in the late batches %.0f" % (c[12] / max(c[8], 1), c[13] / max(c[8], 1), c[14] / max(c[8], 1)))
print("kernel info", g.info())
print("osd kernel ms", pr["osd_ms"], "shots", c[8], "mean pivots", c[9] / max(c[8], 1))
for i, nme in enumerate(names):
    print("%-10s %6.1f %%   %8.0f ticks/shot" % (nme, 100.0 * c[i] / tot, c[i] / max(c[8], 1)))

