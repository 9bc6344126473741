#!/usr/bin/env python3
"""Does the OSD stage of batch i overlap the BP stage of batch i+1 on two streams (two decoders = two workspaces)?
usage (GPU box): tools/overlap_probe.py [shots] [batches]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, helpers
from quits_amd.decoder.device import BatchDecoder, DemSampler, WindowGraph
shots = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, L, pri = helpers.dem_matrices("bb144_custom_r12_p0.003")
smp = DemSampler(H, L, pri)
dets = [smp.sample(shots, seed=5 + i)[0] for i in range(nb)]
g = WindowGraph(H, pri)
decs = [BatchDecoder(g, max_iter=50, osd_method="osd_0") for _ in range(2)]
outs = [(torch.empty((shots, g.words), dtype=torch.int32, device="cuda"), torch.empty((shots,), dtype=torch.int32, device="cuda")) for _ in range(nb)]
for d in decs: d.decode(dets[0])
torch.cuda.synchronize()

def sequential():
    for i in range(nb): decs[0].decode(dets[i], err_bits=outs[i][0], status=outs[i][1])

pr = int(os.environ.get("OSD_PRIO", "0")); s_bp, s_osd = torch.cuda.Stream(), torch.cuda.Stream(priority=pr)
def pipelined():
    evs = []
    for i in range(nb):
        d = decs[i & 1]
        if i >= 2: s_bp.wait_event(evs[i - 2])                 # this decoder's workspace is free again
        d.decode(dets[i], err_bits=outs[i][0], status=outs[i][1], stage=1, stream=s_bp)
        e = torch.cuda.Event(); e.record(s_bp)
        s_osd.wait_event(e)
        d.decode(dets[i], err_bits=outs[i][0], status=outs[i][1], stage=2, stream=s_osd)
        e2 = torch.cuda.Event(); e2.record(s_osd); evs.append(e2)

ref = None
for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    chk = sum(int(o[0].sum().item()) for o in outs)
    if ref is None: ref = chk
    print("%-10s %7.2f ms per batch  %8.0f shots/s   checksum %s" % (name, 1e3 * dt / nb, shots * nb / dt, "same" if chk == ref else "DIFFERENT"))
