#!/usr/bin/env python3
"""Phase breakdown of the rebuilt OSD-CS / OSD-E kernel qd_osdcs_kernel (osd_cs.hip), as wavefront 0 of a workgroup sees it
(needs a -DQD_OSD_TIMING build: tools/build_variants.sh osd, then QUITS_AMD_LIB=build_ablate/lib_osdtiming.so).
  FIXTURE=... WINDOW=W,F,k SHOTS=... OSD_METHOD=osd_cs OSD_ORDER=1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
name = os.environ.get("FIXTURE", "bb144_custom_r12_p0.003")
if os.environ.get("WINDOW"):
    W_, F_, k_ = (int(v) for v in os.environ["WINDOW"].split(","))
    win = helpers.window_set(name, W_, F_)[k_]
    H, pri = win["H"], win["priors"]
    L = H[:8]
else:
    H, L, pri = helpers.dem_matrices(name)
det, obs = DemSampler(H, L, pri).sample(int(os.environ.get("SHOTS", "32768")), seed=5)
g = WindowGraph(H, pri)
d = BatchDecoder(g, max_iter=int(os.environ.get("MAX_ITER", "50")), osd_method=os.environ.get("OSD_METHOD", "osd_cs"), osd_order=int(os.environ.get("OSD_ORDER", "1")))
assert d.info()["post_kernel"] == "qd_osdcs_kernel", d.info()
d.decode(det); torch.cuda.synchronize(); d.debug_counters()
d.set_profiling(True); d.decode(det); torch.cuda.synchronize()
c = d.debug_counters(); pr = d.profile()
names = ["sort (column order)", "[A] push images + clear", "[B] panel pivots (wavefront 0)", "[C] Q update + next scatter", "sweep + output",
         "barrier after A", "barrier after B", "barrier after C"]
shots = max(c[8], 1)
tot = sum(c[:8]) or 1
print("fixture %s  window %s  %d x %d" % (name, os.environ.get("WINDOW", "whole"), H.shape[0], H.shape[1]))
print("osd kernel ms %.2f, shots in OSD %d, mean pivots %.1f, batches per shot %.1f, ticks per shot %.0f (100 MHz)" % (pr["osd_ms"], c[8], c[9] / shots, c[10] / shots, tot / shots))
for i, nme in enumerate(names):
    print("%-34s %6.1f %%   %8.0f ticks/shot   %7.1f ticks/batch" % (nme, 100.0 * c[i] / tot, c[i] / shots, c[i] / max(c[10], 1)))
SUB = {"1": ((11, "[B] liveness sweep"), (12, "[B] chunk load + earlier pivots"), (13, "[B] chunk columns in order"), (14, "[B] write-back + records")),
       "2": ((11, "sort: samples + splitters"), (12, "sort: bucket numbers + counts"), (13, "sort: scan + scatter"), (14, "sort: bucket sorts")),
       "5": ((11, "panel finished -> last barrier: wavefront 1"), (12, "... wavefront 4 (shares wavefront 0's SIMD)"), (13, "... wavefront 2"), (14, "... wavefront 7")),
       "3": ((11, "sweep: residual, Q dump, weight tables"), (12, "sweep: single columns"), (13, "sweep: patterns"), (14, "sweep: winner + output"))}
for i, nme in SUB[os.environ.get("QD_CS_SUB", "1")]:
    print("%-34s %6.1f %%   %8.0f ticks/shot   %7.1f ticks/batch" % (nme, 100.0 * c[i] / tot, c[i] / shots, c[i] / max(c[10], 1)))
