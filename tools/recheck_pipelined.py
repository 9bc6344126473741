#!/usr/bin/env python3
"""The 2 x 10^6 shots of profiles/r02_ler_forms_* decoded once more through the DRIVER (build_circuit_plan -> plan.decode in
2^20-shot calls: the pipelined path, two workspaces, two side streams) and compared, shot by shot, with the stored column of the
oracle's double-precision decoder on the 2^-11 grid.  usage (GPU box): tools/recheck_pipelined.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, helpers
from quits_amd.decoder.base import detector_error_model_to_matrix
from quits_amd.decoder.device import DemSampler
from quits_amd.decoder.sliding_window import build_circuit_plan
from quits_amd.dem import Circuit
NAME, R, MAX_ITER = "bb144_custom_r12_p0.003", 12, 50
circ = Circuit(helpers.circuit_text(NAME))
H, L, pri = helpers.dem_matrices(NAME)
hz = helpers.code("bb144")["hz"]
opts = dict(bp_method="minimum_sum", schedule="parallel", max_iter=MAX_ITER, osd_method="osd_0", osd_order=0)
plan = build_circuit_plan(circ, hz, R + 2, 1, R, dict(opts), dict(opts))
assert plan.pipeline and len(plan.windows) == 1
smp = DemSampler(H, L, pri)
B = 50000                                    # the study's sampling unit (seed, shot0 = c * B)
for seed in (1, 2):
    ref = np.load(os.path.join(ROOT, "profiles", "r02_ler_forms_data", "seed%d_fail_bits.npz" % seed))
    want = np.unpackbits(ref["ldpc_f64_q11_fail"])[:1000000].astype(bool)
    t0 = time.time()
    dets, obss = zip(*[smp.sample(B, seed=seed, shot0=c * B) for c in range(20)])
    det, obs = torch.cat(dets), torch.cat(obss)
    pred = plan.decode(det)                  # 10^6 shots in one call: 16 chunks, pipelined
    fail = (pred != obs).any(dim=1).cpu().numpy()
    same = bool(np.array_equal(fail, want))
    print("seed %d: %d shots through the pipelined driver, %d logical failures; failure bits identical to the oracle's grid column: %s  (%.1f s)"
          % (seed, fail.size, int(fail.sum()), same, time.time() - t0))
    assert same
