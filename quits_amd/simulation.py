"""Monte-Carlo drivers around the decoder, mirroring `/root/reference/src/quits/simulation.py`.

  get_codecap_pL       <- simulation.py:31-61   code-capacity logical error rate of a decoder plug-in
  get_stim_mem_result  <- simulation.py:8-28    detector / observable samples of a memory circuit

`get_codecap_pL` keeps the reference's signature and its random stream (`np.random.seed(seed)` followed by one
`np.random.binomial(1, p, n)` per trial), so a given seed produces the same noise vectors as the reference.  With a plug-in
that decodes batches (`quits_amd.decoder.BpOsdDecoder`: it has `decode_batch`) all trials go through the device decoder in
one call; any other plug-in class runs the reference's per-trial loop on the host.
"""
from __future__ import annotations

import numpy as np


def get_stim_mem_result(circuit, num_trials, seed=-1):
    """Detector and observable samples for a logical-memory circuit (simulation.py:8-28).

    A real `stim.Circuit` is sampled by Stim exactly like the reference does.  Anything else (circuit text or
    `quits_amd.dem.Circuit`; Stim is not a dependency of this package) is sampled at the level of its detector error
    model on the device: independent fault mechanisms with the model's probabilities, the same distribution for Pauli noise,
    not the same random stream.
    """
    if hasattr(circuit, "compile_detector_sampler"):
        sampler = circuit.compile_detector_sampler(seed=seed) if seed >= 0 else circuit.compile_detector_sampler()
        return sampler.sample(shots=num_trials, separate_observables=True)
    from .decoder.base import detector_error_model_to_matrix
    from .decoder.device import DemSampler
    from .dem import Circuit
    if not isinstance(circuit, Circuit):
        circuit = Circuit(str(circuit))
    H, L, priors = detector_error_model_to_matrix(circuit.detector_error_model())
    if seed < 0:
        seed = int(np.random.SeedSequence().entropy % (1 << 62))
    det, obs = DemSampler(H, L, priors).sample(int(num_trials), seed=int(seed))
    return det.cpu().numpy().astype(bool), obs.cpu().numpy().astype(bool)


def get_codecap_pL(code, p, num_trials, decoder, dict, basis='Z', seed=-1, tqdm_on=False):
    """Code-capacity logical error rate (simulation.py:31-61): i.i.d. bit flips with probability `p` on the data qubits,
    one decode of `H e` per trial, failure when the residual error anticommutes with a logical operator."""
    if seed >= 0:
        np.random.seed(seed)
    basis = basis.upper()
    if basis == 'Z':
        parity_check_matrix, logical_codewords = code.hz, code.lz
    elif basis == 'X':
        parity_check_matrix, logical_codewords = code.hx, code.lx
    else:
        raise ValueError("basis must be 'Z' or 'X'")
    bpd = decoder(parity_check_matrix, **dict)
    n = parity_check_matrix.shape[1]
    H = np.asarray(parity_check_matrix.todense() if hasattr(parity_check_matrix, "todense") else parity_check_matrix) % 2
    Lm = np.asarray(logical_codewords.todense() if hasattr(logical_codewords, "todense") else logical_codewords) % 2
    H = H.astype(np.int64); Lm = Lm.astype(np.int64)
    if hasattr(bpd, "decode_batch") and num_trials > 0:
        # same stream as `num_trials` successive binomial(1, p, n) calls of the legacy generator
        noise = np.random.binomial(1, p, (num_trials, n)).astype(np.int64)
        syndromes = (noise @ H.T) % 2
        decoded = np.asarray(bpd.decode_batch(syndromes.astype(np.uint8))).astype(np.int64)
        residual = (decoded + noise) % 2
        num_errors = int(((residual @ Lm.T) % 2).any(axis=1).sum())
        return num_errors / num_trials
    iterator = range(num_trials)
    if tqdm_on:
        from tqdm import tqdm
        iterator = tqdm(iterator)
    num_errors = 0
    for _ in iterator:
        noise = np.random.binomial(1, p, n)
        syndrome = H @ noise % 2
        decoded_error = np.asarray(bpd.decode(syndrome)).astype(np.int64)
        residual_error = (decoded_error + noise) % 2
        if (Lm @ residual_error % 2).any():
            num_errors += 1
    return num_errors / num_trials
