#!/usr/bin/env python3
"""Re-decode the 2 x 10^6 shots of profiles/r02_ler_forms_* with the current library and compare the per-shot logical failure
bits with the stored columns: the oracle's double-precision decoder on the 2^-11 grid must be reproduced shot for shot.
usage (GPU box): tools/recheck_ler_forms.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
for seed in (1, 2):
    out = "/tmp/recheck_seed%d.npz" % seed
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ler_forms.py"), "gpu", "1000000", str(seed), out], stdout=subprocess.DEVNULL)
    z = np.load(out); ref = np.load(os.path.join(ROOT, "profiles", "r02_ler_forms_data", "seed%d_fail_bits.npz" % seed))
    obs = z["obs"]
    for nm in ("k1_grid", "k1g_grid"):
        fail = np.packbits((z[nm + "_pred"] != obs).astype(np.uint8))
        same_oracle = np.array_equal(fail, ref["ldpc_f64_q11_fail"]); same_before = np.array_equal(fail, ref[nm + "_fail"])
        print("seed %d %-8s: %d shots, %d logical failures; failure bits identical to the oracle's grid column: %s; to the stored device column: %s"
              % (seed, nm, obs.size, int(np.unpackbits(fail)[:obs.size].sum()), same_oracle, same_before))
        assert same_oracle and same_before
