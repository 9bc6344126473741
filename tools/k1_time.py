#!/usr/bin/env python3
"""BP stage alone on the headline window: ms per launch of 65536 shots + a checksum of the outputs (A/B of kernel variants:
QUITS_AMD_LIB=<variant.so> python tools/k1_time.py [code] [shots] [max_iter]).  GPU box."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
name = sys.argv[1] if len(sys.argv) > 1 else "bb144_custom_r12_p0.003"
shots = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
mi = int(sys.argv[3]) if len(sys.argv) > 3 else 50
if os.path.exists(os.path.join(helpers.GOLD, "windows", name + ".npz")):
    H, L, pri = helpers.dem_matrices(name)
else:                                                    # any circuit fixture (e.g. bb144_custom_r12_p0.006): own DEM extractor
    from quits_amd.decoder.base import detector_error_model_to_matrix
    from quits_amd.dem import Circuit
    H, L, pri = detector_error_model_to_matrix(Circuit(helpers.circuit_text(name)))
det, obs = DemSampler(H, L, pri).sample(shots, seed=5)
g = WindowGraph(H, pri); d = BatchDecoder(g, max_iter=mi, osd_method="osd_0")
for stage, key in ((1, "bp_ms"), (3, "osd_ms"))[:int(os.environ.get("K1_STAGES", "2"))]:
    bits, st = d.decode(det, stage=stage); torch.cuda.synchronize()
    d.set_profiling(True); d.profile()
    for _ in range(3):
        bits, st = d.decode(det, stage=stage)
    torch.cuda.synchronize()
    pr = d.profile()
    print("%s stage %d: bp %.3f ms  osd %.3f ms per launch   crc bits %08x status %08x   mean iterations %.2f" % (
        os.environ.get("QUITS_AMD_LIB", "default"), stage, pr["bp_ms"] / 3, pr["osd_ms"] / 3,
        zlib.crc32(bits.cpu().numpy().tobytes()), zlib.crc32((st & 0xFFFFF).cpu().numpy().tobytes()), float((st & 0x3FFF).float().mean())))
    d.set_profiling(False)
