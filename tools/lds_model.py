#!/usr/bin/env python3
"""LDS-array cycles of one iteration of the flooding min-sum kernel (bp_kernels.hip) for a window graph and a slot
assignment, by the instruction table of /opt/skills/guides/MI355X_MICROARCH.md (section LDS):
  ds_read_b32   2 groups of 32 lanes, bank = (addr/4) mod 32, one cycle per distinct address on the busiest bank of a group
  ds_read_b128  4 groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), bank = (addr/4) mod 64, 4 banks per lane
  ds_write_b32  as ds_read_b32, floor 4 cycles (2 LDS-array cycles + the VGPR transfer)
Identical addresses inside a group broadcast.  The round-1 counters of the headline window (7820 LDS cycles per
shot-iteration, 3640 of them conflicts: profiles/r01i_pmc_sq_bp_kernel.txt) are what this model is checked against.

  python tools/lds_model.py [fixture name]      # CPU only"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[x + 32 for x in g] for g in G128]


def cyc_b32(addr):          # addr: int array [64], -1 = inactive lane
    tot = 0
    for g0 in (0, 32):
        a = addr[g0:g0 + 32]
        a = np.unique(a[a >= 0])
        if a.size:
            tot += np.bincount((a // 4) % 32, minlength=32).max()
    return tot


def cyc_b128(addr):         # 16-byte aligned addresses
    tot = 0
    for g in G128:
        a = addr[g]
        a = np.unique(a[a >= 0])
        if a.size:
            tot += np.bincount((a // 16) % 16, minlength=16).max()     # a 16-byte access covers one of 16 four-bank groups
    return tot


def iteration_cycles(H, chk_order, bit_order, chk_addr_of_slot=None, bit_addr_of_slot=None):
    """H: csr (m x n).  chk_order / bit_order: node index per slot (lane assignment: wave w owns slots 64w..64w+63).
    *_addr_of_slot: LDS element index of a slot's state (default = the slot).  Returns dict of cycles (ideal, actual)."""
    from scipy.sparse import csr_matrix, csc_matrix
    Hr = csr_matrix(H); Hr.sort_indices(); Hc = csc_matrix(H); Hc.sort_indices()
    m, n = Hr.shape
    ca = np.arange(m) if chk_addr_of_slot is None else np.asarray(chk_addr_of_slot)
    ba = np.arange(n) if bit_addr_of_slot is None else np.asarray(bit_addr_of_slot)
    chk_slot = np.empty(m, int); chk_slot[chk_order] = np.arange(m)
    bit_slot = np.empty(n, int); bit_slot[bit_order] = np.arange(n)
    llr_addr = lambda j: 4 * ba[bit_slot[j]]            # posterior of fault j
    st_addr = lambda i: 16 * ca[chk_slot[i]]            # state of check i
    out = {"check_gather": [0, 0], "bit_gather": [0, 0], "llr_write": [0, 0], "state_rw": [0, 0]}
    rdeg = np.diff(Hr.indptr); cdeg = np.diff(Hc.indptr)
    for w0 in range(0, m, 64):
        nodes = chk_order[w0:w0 + 64]
        kmax = rdeg[nodes].max()
        for k in range(kmax):
            addr = np.full(64, -1)
            for l, i in enumerate(nodes):
                if k < rdeg[i]:
                    addr[l] = llr_addr(Hr.indices[Hr.indptr[i] + k])
            out["check_gather"][0] += 2; out["check_gather"][1] += max(2, cyc_b32(addr))
        sa = np.full(64, -1); sa[:len(nodes)] = [st_addr(i) for i in nodes]
        out["state_rw"][0] += 4 + 13; out["state_rw"][1] += max(4, cyc_b128(sa)) + 13
    for w0 in range(0, n, 64):
        nodes = bit_order[w0:w0 + 64]
        qmax = cdeg[nodes].max()
        for q in range(qmax):
            addr = np.full(64, -1)
            for l, j in enumerate(nodes):
                if q < cdeg[j]:
                    addr[l] = st_addr(Hc.indices[Hc.indptr[j] + q])
            out["bit_gather"][0] += 4; out["bit_gather"][1] += max(4, cyc_b128(addr))
        wa = np.full(64, -1); wa[:len(nodes)] = [llr_addr(j) for j in nodes]
        out["llr_write"][0] += 4; out["llr_write"][1] += max(4, cyc_b32(wa))
    out["total"] = [sum(v[0] for v in out.values()), sum(v[1] for v in out.values())]
    return out


def host_order(H):
    """The slot assignment of qd_graph_create: degree-descending, stable."""
    from scipy.sparse import csr_matrix, csc_matrix
    rdeg = np.diff(csr_matrix(H).indptr); cdeg = np.diff(csc_matrix(H).indptr)
    return np.argsort(-rdeg, kind="stable"), np.argsort(-cdeg, kind="stable")


if __name__ == "__main__":
    import helpers
    name = sys.argv[1] if len(sys.argv) > 1 else "bb144_custom_r12_p0.003"
    H, L, pri = helpers.dem_matrices(name)
    co, bo = host_order(H)
    r = iteration_cycles(H, co, bo)
    for k, v in r.items():
        print("%-14s ideal %6d   with conflicts %6d   (+%.0f %%)" % (k, v[0], v[1], 100.0 * (v[1] - v[0]) / max(1, v[0])))
