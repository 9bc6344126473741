"""Import the read-only reference (`/root/reference/src/quits`) in THIS container.

`stim` and `ldpc` are not installed here, so `import quits` fails on
`decoder/base.py:9` and `decoder/bposd.py:5`.  The only Stim use on the code/circuit
construction side is the final `stim.Circuit(text)` wrap (`qldpc_code/bb.py:301`,
`circuit_construction/cardinal.py:267`), so a `str`-subclass stub is enough to make the
reference emit the circuit *text* unmodified.  `ldpc` is replaced by dummy classes; nothing
in the fixtures calls them.

Only used by tools/gen_fixtures.py; never imported by the package, the tests or bench.py
(`/root/reference` does not exist on the GPU box).
"""
import sys
import types

REFERENCE_SRC = "/root/reference/src"


class _StubCircuit(str):
    """Stands in for stim.Circuit: keeps the program text the reference built."""


def import_reference():
    sys.dont_write_bytecode = True  # never create __pycache__ under /root/reference
    if "stim" not in sys.modules:
        stim = types.ModuleType("stim")
        stim.Circuit = _StubCircuit
        stim.DetectorErrorModel = object
        stim.DemTarget = object
        sys.modules["stim"] = stim
    if "ldpc" not in sys.modules:
        ldpc = types.ModuleType("ldpc")
        for sub, cls in (("bposd_decoder", "BpOsdDecoder"), ("bplsd_decoder", "BpLsdDecoder")):
            m = types.ModuleType("ldpc." + sub)
            setattr(m, cls, type(cls, (), {}))
            setattr(ldpc, sub, m)
            sys.modules["ldpc." + sub] = m
        sys.modules["ldpc"] = ldpc
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import quits  # noqa: F401

    return quits
