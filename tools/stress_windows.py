#!/usr/bin/env python3
"""Every (W, F) window shape for one circuit: batched device driver vs the oracle's C restatement of the reference loop
(sliding_window.py:104-188), circuit-level and phenomenological, identical logical predictions.  GPU box.
usage: tools/stress_windows.py [shots]"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import helpers, oracle as orc
from quits_amd.decoder import sliding_window_bposd_circuit_mem, sliding_window_bposd_phenom_mem
from quits_amd.decoder.base import spacetime, window_count
from quits_amd.decoder.sliding_window import phenom_window_matrices
from quits_amd.dem import Circuit


def run(shots=96, name="bb72_custom_r6_p0.003", code="bb72", R=6, opts=(("minimum_sum", "parallel", 12, "osd_0", 0),
                                                                          ("product_sum", "serial", 2, "osd_cs", 2))):
    cd = helpers.code(code)
    hz, lz = cd["hz"], cd["lz"]
    nz = hz.shape[0]
    circ = Circuit(helpers.circuit_text(name))
    H, L, pri = helpers.dem_matrices(name)
    det, obs, _ = orc.sample_dem(H, L, pri, seed=77, shot0=0, B=shots)
    t0 = time.time(); n_ok = 0
    for W in range(1, R + 4):
        for F in range(1, W + 1):
            for (method, sched, mi, osd, order) in opts:
                grid = (method, sched) == ("minimum_sum", "parallel")          # exact arithmetic on the LLR grid <-> double form
                prm = orc.make_params(method, sched, mi, osd, order, 1.0, orc.FORM_LDPC_F64 if grid else orc.FORM_LDPC_F32)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    pred = sliding_window_bposd_circuit_mem(det, circ, hz, lz, W, F, max_iter=mi, osd_order=order,
                                                            bp_method=method, schedule=sched, osd_method=osd)
                    ncr, _, _ = window_count(R, W, F)
                    checks, commits, priors, updates = spacetime(circ, hz, W, F, ncr)
                wins = [{"H": checks[k], "L": commits[k], "priors": priors[k], "U": updates[k] if k < ncr else None,
                         "row0": F * k * nz} for k in range(len(checks))]
                ref, _ = orc.sliding_window_decode(wins, nz, det, prm, device_grid=grid)
                assert np.array_equal(pred, ref.astype(np.int64)), ("circuit", W, F, method, sched)
                n_ok += 1
    print("window sweep: %d (W, F, options) combinations identical, %.0f s" % (n_ok, time.time() - t0))
    return n_ok


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 96)
