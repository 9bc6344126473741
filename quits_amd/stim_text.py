"""Parser for the Stim circuit dialect that QUITS emits.

The reference builds its circuits as *text* (`/root/reference/src/quits/circuit.py:30-279`) and
only wraps the text in `stim.Circuit` at the very end (`qldpc_code/bb.py:301`,
`circuit_construction/cardinal.py:267`, `zxcoloration.py:270`).  Stim is not installed in the
build image, so the decoder front end reads the same text itself.

Ops understood: R RX H CX M MX MR TICK X_ERROR Z_ERROR DEPOLARIZE1
DEPOLARIZE2 DETECTOR OBSERVABLE_INCLUDE REPEAT{...} (one level; nesting is accepted too).
PAULI_CHANNEL_1/2 (emitted only when an ErrorModel field is a tuple, `circuit.py:115,139,171`)
need Stim's approximate-disjoint-error conversion and are rejected with a clear error.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

_GATES = {"R", "RX", "H", "CX", "M", "MX", "MR"}
_NOISE = {"X_ERROR", "Z_ERROR", "DEPOLARIZE1", "DEPOLARIZE2"}
_ANNOT = {"DETECTOR", "OBSERVABLE_INCLUDE"}
_IGNORED = {"TICK", "QUBIT_COORDS", "SHIFT_COORDS"}
_COORD_ARGS = {"DETECTOR", "QUBIT_COORDS", "SHIFT_COORDS"}     # parenthesised arguments are coordinates: ignored
_UNSUPPORTED = {"PAULI_CHANNEL_1", "PAULI_CHANNEL_2", "Y_ERROR", "CORRELATED_ERROR", "E",
                "ELSE_CORRELATED_ERROR", "MPP", "MY", "RY", "MRX", "MRY", "CZ", "CY", "SWAP",
                "S", "S_DAG", "SQRT_X", "X", "Y", "Z", "CNOT", "ZCX"}


@dataclass(frozen=True)
class Op:
    """One flattened circuit instruction.

    name    : gate / noise / annotation mnemonic
    arg     : parenthesised argument (probability, or observable index), 0.0 if none
    targets : qubit indices, or (for DETECTOR / OBSERVABLE_INCLUDE) *absolute* measurement indices
    """
    name: str
    arg: float
    targets: Tuple[int, ...]


class CircuitSyntaxError(ValueError):
    pass


def _parse_line(line: str):
    head, _, rest = line.partition(" ")
    arg = 0.0
    if "(" in head or (rest.startswith("(")):
        # e.g. "X_ERROR(0.003) 1 2"  (the reference never puts a space before '(')
        name, _, tail = line.partition("(")
        argtxt, _, rest = tail.partition(")")
        name = name.strip()
        if name.upper() in _COORD_ARGS:
            # DETECTOR(x, y, t) / QUBIT_COORDS(x, y) / SHIFT_COORDS(...): coordinates carry no decoding information
            return name.upper(), 0.0, rest.split()
        if "," in argtxt:
            raise NotImplementedError(
                f"{name} with a multi-parameter channel is not supported (reference emits it only for "
                "tuple-valued ErrorModel fields, circuit.py:115,139,171); use scalar rates")
        arg = float(argtxt)
    else:
        name = head
    return name.strip().upper(), arg, rest.split()


def flatten(text: str) -> Tuple[List[Op], int, int, int]:
    """Expand REPEAT blocks and resolve rec[-k] to absolute measurement indices.

    Returns (ops, num_measurements, num_detectors, num_observables).
    """
    lines = [ln.split("#", 1)[0].strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln]
    ops: List[Op] = []
    state = {"meas": 0, "det": 0, "obs": 0}

    def run(block: List[str]):
        i = 0
        while i < len(block):
            ln = block[i]
            if ln.upper().startswith("REPEAT"):
                parts = ln.replace("{", " { ").split()
                if len(parts) < 3 or parts[2] != "{":
                    raise CircuitSyntaxError(f"malformed REPEAT header: {ln!r}")
                count = int(parts[1])
                depth, j = 1, i + 1
                while j < len(block):
                    if block[j].upper().startswith("REPEAT"):
                        depth += 1
                    elif block[j] == "}":
                        depth -= 1
                        if depth == 0:
                            break
                    j += 1
                if depth != 0:
                    raise CircuitSyntaxError("unterminated REPEAT block")
                body = block[i + 1:j]
                for _ in range(count):
                    run(body)
                i = j + 1
                continue
            if ln == "}":
                raise CircuitSyntaxError("unbalanced '}'")
            name, arg, toks = _parse_line(ln)
            if name in _IGNORED:
                pass
            elif name in _GATES or name in _NOISE:
                try:
                    qs = tuple(int(t) for t in toks)
                except ValueError as exc:
                    raise CircuitSyntaxError(f"bad qubit target in {ln!r}") from exc
                if name in ("M", "MX", "MR") and arg != 0.0:
                    # Stim's measurement-flip noise M(p): the reference models measurement errors with X_ERROR / Z_ERROR
                    # before the measurement instead (circuit.py:201-246); dropping the argument would silently empty the DEM
                    raise NotImplementedError(f"{name}({arg:g}): noisy-measurement arguments are outside the QUITS dialect "
                                              "handled here (the reference emits X_ERROR/Z_ERROR before M/MX/MR)")
                if name in ("CX", "DEPOLARIZE2") and len(qs) % 2:
                    raise CircuitSyntaxError(f"{name} needs an even number of targets")
                ops.append(Op(name, arg, qs))
                if name in ("M", "MX", "MR"):
                    state["meas"] += len(qs)
            elif name in _ANNOT:
                recs = []
                for t in toks:
                    if not (t.startswith("rec[-") and t.endswith("]")):
                        raise CircuitSyntaxError(f"unsupported target {t!r} in {ln!r}")
                    k = int(t[5:-1])
                    idx = state["meas"] - k
                    if idx < 0:
                        raise CircuitSyntaxError(f"{t} looks before the first measurement")
                    recs.append(idx)
                if name == "DETECTOR":
                    ops.append(Op(name, float(state["det"]), tuple(recs)))
                    state["det"] += 1
                else:
                    obs = int(arg)
                    ops.append(Op(name, float(obs), tuple(recs)))
                    state["obs"] = max(state["obs"], obs + 1)
            elif name in _UNSUPPORTED:
                raise NotImplementedError(f"Stim instruction {name} is outside the QUITS dialect handled here")
            else:
                raise CircuitSyntaxError(f"unknown instruction {name!r}")
            i += 1

    run(lines)
    return ops, state["meas"], state["det"], state["obs"]
