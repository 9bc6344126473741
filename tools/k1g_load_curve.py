#!/usr/bin/env python3
"""BP stage of the per-edge kernel (qd_bp_edge_kernel: product_sum / serial, the reference's defaults) on one W = 5 window of the
[[144,12,12]] circuit: ms per launch against the number of shots in the launch and against max_iter.  GPU box.
    python tools/k1g_load_curve.py [bp_method] [schedule]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
bp_method = sys.argv[1] if len(sys.argv) > 1 else "product_sum"
schedule = sys.argv[2] if len(sys.argv) > 2 else "serial"
name = os.environ.get("FIXTURE", "bb144_custom_r12_p0.003")
W_, F_, k_ = (int(v) for v in os.environ.get("WINDOW", "5,3,1").split(","))
win = helpers.window_set(name, W_, F_)[k_]
H, pri = win["H"], win["priors"]
L = H[:8]
g = WindowGraph(H, pri)
print("window", H.shape, "nnz", H.nnz, g.info())
shots_list = [int(v) for v in os.environ.get("SHOTS", "16384,32768,49152,65536,81920,98304,131072,163840").split(",")]
det_all, _ = DemSampler(H, L, pri).sample(max(shots_list), seed=5)
for mi in [int(v) for v in os.environ.get("MAX_ITER", "10,3,1").split(",")]:
    d = BatchDecoder(g, max_iter=mi, bp_method=bp_method, schedule=schedule, osd_method="osd_cs", osd_order=1)
    d.reserve(max(shots_list))
    for B in shots_list:
        det = det_all[:B]
        bits, st = d.decode(det, stage=1); torch.cuda.synchronize()
        d.set_profiling(True); d.profile()
        for _ in range(3):
            bits, st = d.decode(det, stage=1)
        torch.cuda.synchronize()
        pr = d.profile()
        d.set_profiling(False)
        it = (st & 0x3FFF).float()
        print("max_iter %2d  shots %6d  (%.2f workgroups of 64 per CU)  bp %.3f ms per launch  %.1f k shots/s  mean iterations %.2f  converged %.3f  crc %08x" % (
            mi, B, B / 64 / 256, pr["bp_ms"] / 3, B / (pr["bp_ms"] / 3), float(it.mean()), float(((st >> 16) & 1).float().mean()),
            zlib.crc32(bits.cpu().numpy().tobytes())))
    d.release_workspace()
