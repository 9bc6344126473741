"""Sliding-window decoders (S. Huang and S. Puri, PRA 110, 012453) with the `quits.decoder` call surface.

Mirrors `/root/reference/src/quits/decoder/sliding_window.py`:
  sliding_window_phenom_mem   <- :14-101
  sliding_window_circuit_mem  <- :104-188
Same positional order, keyword names, return type (int64 [shots, logicals]) and errors.

Two execution paths, chosen by the plug-in decoder class, never silently by availability:
  * decoder classes are `quits_amd.decoder.BpOsdDecoder` (the HIP decoder): the loops are inverted -- windows outer,
    the whole shot batch inner -- and everything between the detector record and the logical prediction stays on
    the MI355X (the reference's per-shot Python loop tops out at a few thousand shots/s on its own);
  * any other class (e.g. ldpc's, or the CPU oracle in the tests): the reference's per-shot loop, restated, so
    third-party plug-ins keep working exactly as upstream.
"""
from __future__ import annotations

import warnings

import numpy as np
from scipy.sparse import csc_matrix, csr_matrix

from .base import spacetime, window_count

_CHUNK = 1 << 16   # shots per device batch
_CHUNK_EDGE = 5 << 14   # ... when BP runs in the one-message-per-edge kernel: five 64-shot workgroups per CU are resident on 256 CUs and the kernel's
                        # time is per-workgroup latency (81 920 vs 65 536 shots per launch: +8 % shots/s, 98 304 -18 %; profiles/r03o)


def _env_int(name, default):
    import os
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def _env_flag(name):
    """True for any non-empty value other than '0' / 'false' / 'no' / 'off' (so QD_NO_PIPELINE=true still means what it says)."""
    import os
    v = os.environ.get(name, "").strip().lower()
    return v not in ("", "0", "false", "no", "off")


def _current_device():
    """Index of the current CUDA device (-1 without a GPU): plans, graphs, decoders and workspaces are bound to it."""
    try:
        import torch
        return int(torch.cuda.current_device()) if torch.cuda.is_available() else -1
    except Exception:            # pragma: no cover
        return -1


def _progress(it, on):
    if on:
        try:
            from tqdm import tqdm
            return tqdm(it)
        except ImportError:  # pragma: no cover
            return it
    return it


def phenom_window_matrices(hz, W, F, W_last):
    """Analytic window matrices of the phenomenological variant (reference sliding_window.py:56-68) plus the
    commit/hand-off selectors the reference applies by slicing (:86,:88,:96): for window matrix
    H_w = [ I_W (x) hz | B (x) I_nz ],  data block f of the decoded vector is e[f*nq:(f+1)*nq] and the syndrome update
    is the measurement block F-1."""
    hz = np.asarray(hz) % 2
    nz, nq = hz.shape
    B = np.eye(W, dtype=int)
    for i in range(1, W):
        B[i, i - 1] = 1
    h_mid = np.column_stack((np.kron(np.eye(W, dtype=int), hz), np.kron(B, np.eye(nz, dtype=int))))
    B_last = np.eye(W_last, dtype=int)
    for i in range(1, W_last):
        B_last[i, i - 1] = 1
    B_last = B_last[:, :W_last - 1]
    h_last = np.column_stack((np.kron(np.eye(W_last, dtype=int), hz), np.kron(B_last, np.eye(nz, dtype=int))))
    return csc_matrix(h_mid), csc_matrix(h_last)


def _is_device_decoder(cls) -> bool:
    from .bposd import BpOsdDecoder
    return isinstance(cls, type) and issubclass(cls, BpOsdDecoder)


def _to_device_samples(zcheck_samples):
    import torch
    if isinstance(zcheck_samples, torch.Tensor):
        t = zcheck_samples
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)
        elif t.dtype != torch.uint8:
            t = torch.remainder(t, 2).to(torch.uint8)
        return t.to("cuda").contiguous()
    a = np.asarray(zcheck_samples)
    if a.dtype != np.uint8:
        a = (a % 2).astype(np.uint8) if a.dtype != np.bool_ else a.astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


class DeviceWindowPlan:
    """Everything the batched driver needs, resident on the GPU: per window a decoder, the commit matrix L_k, the
    hand-off matrix U_k and the first detector row."""

    def __init__(self, checks, commits, priors, updates, row0, nz, nobs, dict1, dict2):
        from .device import BatchDecoder, GF2Matrix, WindowGraph
        self.nz, self.nobs = int(nz), int(nobs)
        self.windows = []
        nwin = len(checks)
        cache = {}
        for k in range(nwin):
            kw = dict(dict2 if k == nwin - 1 else dict1)
            kw.pop("error_rate", None)
            kw.pop("channel_probs", None)
            kw.pop("error_channel", None)
            key = (id(checks[k]), id(priors[k]), k == nwin - 1)    # the phenomenological windows share one matrix
            if key not in cache:
                graph = WindowGraph(checks[k], priors[k])
                cache[key] = (graph, BatchDecoder(graph, **kw))
            graph, dec = cache[key]
            Lk = csr_matrix(commits[k])
            if Lk.shape[1] < graph.n:     # L_k only spans the committed columns (base.py:170)
                Lk = csr_matrix((Lk.data, Lk.indices, Lk.indptr), shape=(Lk.shape[0], graph.n))
            U = None
            if k < nwin - 1:
                Uk = csr_matrix(updates[k])
                Uk = csr_matrix((Uk.data, Uk.indices, Uk.indptr), shape=(Uk.shape[0], graph.n))
                U = GF2Matrix(Uk)
            self.windows.append({"dec": dec, "graph": graph, "L": GF2Matrix(Lk), "U": U, "row0": int(row0[k]), "H": checks[k], "kw": kw})
        # the per-edge BP kernel (product_sum / serial) keeps its messages in HBM, one workspace per decoder: split a fixed
        # budget among the windows' decoders instead of letting each claim the single-decoder default
        decs = self.decoders()
        self.chunk = _CHUNK_EDGE if any(d.info()["edge_kernel"] for d in decs) else _CHUNK
        import os
        if os.environ.get("QD_CHUNK_SHOTS"):                      # A/B switch (profiles/r03x_chunk_size_pipelined_ab.txt)
            self.chunk = int(os.environ["QD_CHUNK_SHOTS"])
        if len(decs) > 1:
            budget = float(os.environ.get("QD_GENERAL_WS_GB", "96")) * (1 << 30)
            for d in decs:
                d.set_workspace_limit(max(1 << 28, int(budget / len(decs))))
        # calls of two or more chunks: the post-processing of one chunk beside the BP of another (_decode_pipelined_impl);
        # QD_NO_PIPELINE=1 or plan.pipeline = False keeps everything on the caller's stream.  Not the default where BP runs in the
        # per-edge kernel: HBM-bound, it loses more to the co-running post-processor than the overlap returns (W = 5 / F = 3
        # windows with the reference's settings 219 k -> 209 k shots/s, profiles/r03x_pipelined_driver_multiwindow_ab.txt)
        self.pipeline = not _env_flag("QD_NO_PIPELINE") and (_env_flag("QD_PIPELINE_EDGE") or not any(d.info()["edge_kernel"] for d in decs))
        # lanes of the pipelined driver: 2; 3 for plans of several windows -- a lane's next BP stage waits for its last post stage, which
        # runs beside the BP of the NEXT lane and, starved of wavefront slots by it, ends ~0.4 ms after it: with two lanes the BP stream
        # waits that long before every stage, with three the post stage has one more BP stage's time (profiles/r06_three_lanes_ab.txt)
        self.lanes = max(2, _env_int("QD_PIPELINE_LANES", 3 if nwin > 1 else 2))
        # every lane has its own decoders and every decoder its own posterior workspace (4 bytes per fault and shot of a chunk): a plan of many
        # large windows -- QLP [[1020,136]] W = 3: 18 decoders x 18 900 faults = 1.36 MB per shot and lane -- would not fit three lanes of
        # 65 536 shots (268 GB).  Lanes first, then the chunk, give way until the estimate fits QD_POST_WS_GB (default 160 of the 288 GB).
        if self.pipeline and not os.environ.get("QD_CHUNK_SHOTS"):
            per_shot = sum(4 * ((d.graph.n + 63) // 64 * 64) + 64 for d in decs)
            self.lanes, self.chunk = fit_lanes_and_chunk(per_shot, self.lanes, self.chunk, float(os.environ.get("QD_POST_WS_GB", "160")) * (1 << 30))
        self._side = None
        self._stage = None
        self.host_piece = max(1, _env_int("QD_HOST_PIECE_SHOTS", self.lanes * self.chunk))   # shots per staged piece (one group of the pipelined driver's lanes)
        self.device = _current_device()      # graphs, decoders and workspaces were created on this device
        import threading
        self._lock = threading.RLock()       # one decode_host at a time per plan (staging buffers, side streams and workspaces are per plan);
                                             # re-entrant, and the plan cache takes it (non-blocking) before releasing an idle plan's workspaces

    def release_workspaces(self):
        """Hand the decoders' device workspaces and the staging buffers back (the plan itself -- graphs, decoders, matrices --
        stays): cached plans that are not the one in use hold no large allocations."""
        for w in self.windows:
            for d in w.get("lane_decs", [w["dec"]]):
                d.release_workspace()
        self._stage = None

    def window_matrices(self):
        """Host copies of the window check matrices, in window order (bench.py derives its work model from them)."""
        return [w["H"] for w in self.windows]

    def decoders(self):
        out = []
        for w in self.windows:
            if w["dec"] not in out:
                out.append(w["dec"])
        return out

    def decode(self, det, stats=None):
        """det: cuda uint8 [N, ndet]  ->  cuda uint8 [N, nobs] logical predictions.

        Chunks of `self.chunk` shots, windows inner.  `stats`, if given, receives (window index, status tensor) pairs.
        A call of two or more chunks runs a chunk's post-processing on a second stream beside the BP of the other chunk of its
        pair (_decode_pipelined_impl: headline 1.21 -> 1.30 M shots/s, BP-LSD order 1 690 k -> 866 k, p = 6e-3 320 k -> 355 k,
        W = 3 / F = 1 windows 511 k -> 539 k, identical outputs; profiles/r03x_pipelined_driver_ab.txt,
        r03x_pipelined_driver_multiwindow_ab.txt); results are delivered in order on the caller's stream.  QD_NO_PIPELINE=1 or
        `plan.pipeline = False` turns it off; it is off by default for plans whose BP runs in the per-edge kernel."""
        import torch
        N = det.shape[0]
        if self.pipeline and N >= 2 * self.chunk:
            return self._decode_pipelined(det, stats)
        pred = torch.zeros((N, self.nobs), dtype=torch.uint8, device=det.device)
        for c0 in range(0, N, self.chunk):
            chunk = det[c0:c0 + self.chunk]
            acc = pred[c0:c0 + self.chunk]
            upd = None
            for k, w in enumerate(self.windows):
                err_bits, status = w["dec"].decode(chunk, w["row0"], upd)
                w["L"].xor_apply(err_bits, acc, accumulate=True)
                if w["U"] is not None:
                    upd = torch.empty((chunk.shape[0], self.nz), dtype=torch.uint8, device=det.device)
                    w["U"].xor_apply(err_bits, upd, accumulate=False)
                if stats is not None:
                    stats.append((k, status))
        return pred

    def _decode_pipelined(self, det, stats=None):
        return _decode_pipelined_impl(self, det, stats)

    def decode_host(self, zcheck_samples):
        """The reference call's data path: host samples [N, ndet] (bool / uint8 / any integer numpy array, or a torch tensor)
        -> int64 numpy [N, nobs] (reference sliding_window.py:160,186).  Host arrays are streamed: pieces of `self.host_piece`
        shots go through two pinned staging buffers and a copy stream, so that the host-side copy and the PCIe transfer of one
        piece run beside the decoding of the previous one, and the predictions come back through a pinned buffer.  Tensors that
        already live on the GPU skip the staging."""
        import torch
        if self.device >= 0 and torch.cuda.current_device() != self.device:
            raise RuntimeError("this plan was built on cuda:%d but the current device is cuda:%d (plans are per device; the plan "
                               "cache keys on the current device)" % (self.device, torch.cuda.current_device()))
        with self._lock:
            self._ws_live = True                 # (a concurrent cache lookup may have released them since this plan was handed out)
            return self._decode_host_locked(zcheck_samples)

    def _decode_host_locked(self, zcheck_samples):
        import torch
        dev = torch.device("cuda", self.device) if self.device >= 0 else torch.device("cuda")
        if isinstance(zcheck_samples, torch.Tensor) and zcheck_samples.is_cuda:
            if zcheck_samples.device != dev:
                raise RuntimeError("samples live on %s, the plan on %s" % (zcheck_samples.device, dev))
            return self.decode(_to_device_samples(zcheck_samples)).cpu().numpy().astype(np.int64)
        a = zcheck_samples.cpu().numpy() if isinstance(zcheck_samples, torch.Tensor) else np.asarray(zcheck_samples)
        if a.ndim != 2:
            raise ValueError("zcheck_samples must be a [shots, detectors] array")
        N, ndet = a.shape
        if N == 0:
            return np.zeros((0, self.nobs), dtype=np.int64)
        as_u8 = (lambda x: x.view(np.uint8)) if a.dtype == np.bool_ else ((lambda x: x) if a.dtype == np.uint8 else (lambda x: (x % 2).astype(np.uint8)))
        piece = max(self.chunk, int(self.host_piece) // self.chunk * self.chunk)
        if N <= self.chunk:
            piece = N
        st = self._stage
        if st is None or st["ndet"] != ndet or st["piece"] < min(piece, N):
            rows = min(piece, N)
            st = {"ndet": ndet, "piece": rows,
                  "pin": [torch.empty((rows, ndet), dtype=torch.uint8, pin_memory=True) for _ in range(2)],
                  "dev": [torch.empty((rows, ndet), dtype=torch.uint8, device=dev) for _ in range(2)],
                  "copy": torch.cuda.Stream(device=dev), "out": None}
            self._stage = st
        if st["out"] is None or st["out"].shape[0] < N:
            st["out"] = torch.empty((N, self.nobs), dtype=torch.uint8, pin_memory=True)
        out = st["out"]
        cur = torch.cuda.current_stream()
        h2d_done, dec_done = [None, None], [None, None]
        # calls of two or more chunks: the pieces go through the pipelined driver as ONE chain (round 6): the two lanes carry over from piece to piece, a piece's
        # BP starts behind its own host-to-device copy, its predictions leave on the post stream; the caller's stream only waits at the very end
        chained = self.pipeline and not _env_flag("QD_NO_HOST_CHAIN") and N >= 2 * self.chunk
        chain = {} if chained else None
        res = np.empty((N, self.nobs), dtype=np.int64)
        span = [None, None]                                        # the rows of `out` that the piece in flight on a lane will fill
        try:
            for i, lo in enumerate(range(0, N, st["piece"])):
                hi = min(N, lo + st["piece"])
                b = i & 1
                if h2d_done[b] is not None:
                    h2d_done[b].synchronize()                      # the staging buffer has left for the GPU
                if dec_done[b] is not None:
                    dec_done[b].synchronize()                      # the device buffer has been decoded, its predictions are in `out`:
                    res[span[b][0]:span[b][1]] = out[span[b][0]:span[b][1]].numpy()     # widened here, beside the decoding of the next piece
                span[b] = (lo, hi)
                ready = []
                with torch.cuda.stream(st["copy"]):                # chunk by chunk: the first chunk's BP starts behind ITS copy, not the piece's
                    for c0 in range(0, hi - lo, self.chunk):
                        c1 = min(hi - lo, c0 + self.chunk)
                        np.copyto(st["pin"][b][c0:c1].numpy(), as_u8(a[lo + c0:lo + c1]))
                        st["dev"][b][c0:c1].copy_(st["pin"][b][c0:c1], non_blocking=True)
                        ready.append(torch.cuda.Event())
                        ready[-1].record(st["copy"])
                h2d_done[b] = ready[-1]
                if chained:                                        # (a ragged last piece too, whatever its size: it runs beside the piece before it)
                    chain["ready"] = ready
                    pred = _decode_pipelined_impl(self, st["dev"][b][:hi - lo], None, chain)
                    with torch.cuda.stream(chain["s_post"]):       # (in order behind the piece's post stages)
                        out[lo:hi].copy_(pred, non_blocking=True)
                        dec_done[b] = torch.cuda.Event()
                        dec_done[b].record(chain["s_post"])
                else:
                    cur.wait_event(h2d_done[b])
                    pred = self.decode(st["dev"][b][:hi - lo])
                    out[lo:hi].copy_(pred, non_blocking=True)
                    dec_done[b] = torch.cuda.Event()
                    dec_done[b].record(cur)
        finally:
            if chain:
                for key in ("s_bp", "s_post"):
                    if key in chain:
                        chain[key].synchronize()
            cur.synchronize()
            st["copy"].synchronize()
        for b in (0, 1):
            if span[b] is not None:
                res[span[b][0]:span[b][1]] = out[span[b][0]:span[b][1]].numpy()
        return res


def fit_lanes_and_chunk(per_shot_bytes, lanes, chunk, budget_bytes, min_chunk=8192):
    """(lanes, chunk) of the pipelined driver such that lanes x chunk x per_shot_bytes (the lanes' decoder workspaces) fits the budget: a third
    lane goes first (it is worth 2-5 %), then the chunk is halved (down to `min_chunk` shots: below that the launches are all tail)."""
    while lanes * chunk * per_shot_bytes > budget_bytes:
        if lanes > 2:
            lanes -= 1
        elif chunk // 2 >= min_chunk:
            chunk //= 2
        else:
            break
    return lanes, chunk


def lane_groups(nchunks, lanes):
    """Chunks of a call dealt to groups of at most `lanes`, as few groups as possible and as even as they come: 4 chunks on three lanes are 2 + 2,
    not 3 + 1 (a chunk alone overlaps nothing), 16 on three are 3 + 3 + 3 + 3 + 2 + 2."""
    if nchunks <= 0:
        return []
    ngrp = (nchunks + lanes - 1) // lanes
    return [nchunks // ngrp + (1 if g < nchunks % ngrp else 0) for g in range(ngrp)]


def _decode_pipelined_impl(plan, det, stats, chain=None):
    """Calls of two or more chunks: the BP stages run on one side stream, the post-processing (OSD / LSD over the shots BP
    parked, acc ^= L e, the hand-off U e) on a second one, so that a chunk's post-processing runs beside the BP of the other
    chunk of its pair -- the post-processors are chains of dependent steps that leave most issue slots of a CU idle, BP fills
    them (profiles/r03x_overlap_probe.txt, r03x_pipelined_driver_ab.txt).  Chunks are taken two at a time (lanes 0 / 1, each
    with its own set of decoders = workspaces), windows outer inside a pair:

        BP stream:    BP(A, 0)  BP(B, 0)    BP(A, 1)    BP(B, 1)   ...
        post stream:            post(A, 0)  post(B, 0)  post(A, 1) ...

    BP(X, k) waits for post(X, k - 1) (its syndrome needs that hand-off; it also frees the lane's buffers and decoder), post(X, k)
    for BP(X, k); both streams are in order.  Both start after everything queued on the caller's stream so far (inputs, the
    zeroed accumulator); the caller's stream resumes after the last post stage, which by stream order is after all the others.
    Every buffer is allocated on the caller's stream before the side streams start and none is released before that point.

    `chain` (decode_host, round 6): a dict that carries the two lanes from one call to the next -- the lane buffers, the lanes' last post-stage
    events, every buffer handed out -- so that the BP stream of piece i + 1 starts behind its INPUT (chain["ready"], one event per chunk of the
    piece's host-to-device copy) and the lanes, not behind piece i's last post stage: the caller's stream is not made to wait at all, the caller synchronises the side
    streams itself when it has queued everything (profiles/r06_host_chain_ab.txt)."""
    import torch
    from .device import BatchDecoder
    NL = int(plan.lanes)
    if any(len(w.get("lane_decs", ())) < NL for w in plan.windows):
        import os
        by_first = {}                                 # windows that share a decoder share its lanes' decoders (each with its own workspaces)
        for w in plan.windows:
            lst = by_first.setdefault(id(w["dec"]), list(w.get("lane_decs", [w["dec"]])))
            while len(lst) < NL:
                lst.append(BatchDecoder(w["graph"], **w["kw"]))
            w["lane_decs"] = lst
        if any(d.info()["edge_kernel"] for d in plan.decoders()):
            budget = float(os.environ.get("QD_GENERAL_WS_GB", "96")) * (1 << 30)
            every = [d for lst in by_first.values() for d in lst]
            for d in every:
                d.set_workspace_limit(max(1 << 28, int(budget / len(every))))
    if plan._side is None:
        # (a high-priority post-processing stream, QD_POST_STREAM_PRIORITY=-1, measured no different: profiles/r03x_post_stream_priority_ab.txt)
        plan._side = {}
    if det.device not in plan._side:              # (streams live on the device of the data, one pair per device)
        plan._side[det.device] = (torch.cuda.Stream(device=det.device),
                                  torch.cuda.Stream(device=det.device, priority=_env_int("QD_POST_STREAM_PRIORITY", 0)))
    s_bp, s_post = plan._side[det.device]
    N, C, nwin = det.shape[0], plan.chunk, len(plan.windows)
    dev = det.device
    cur = torch.cuda.current_stream()
    pred = torch.zeros((N, plan.nobs), dtype=torch.uint8, device=dev)
    words = max(w["graph"].words for w in plan.windows)
    if chain is not None and "err_l" in chain:
        err_l, upd_l, st_all = chain["err_l"], chain["upd_l"], chain["st_all"]
    else:
        err_l = [torch.empty((C * words,), dtype=torch.int32, device=dev) for _ in range(NL)]
        upd_l = [torch.empty((C, plan.nz), dtype=torch.uint8, device=dev) for _ in range(NL)] if nwin > 1 else [None] * NL
        st_all = torch.empty((nwin if stats is not None else 1, (N if stats is not None else NL * C)), dtype=torch.int32, device=dev)
        if chain is not None:
            chain.update(err_l=err_l, upd_l=upd_l, st_all=st_all)
    start = torch.cuda.Event()
    start.record(cur)                      # (the zeroed accumulator, the buffers)
    s_bp.wait_event(start)
    s_post.wait_event(start)
    if chain is not None:
        chain.setdefault("keep", []).append(pred)              # nothing handed out may go back to the allocator before the caller has synchronised
    post_done = list(chain.get("post_done", [None] * NL)) if chain is not None else [None] * NL
    # chunks are taken in groups of up to NL, the groups as even as they come (4 chunks on three lanes: 2 + 2, not 3 + 1 -- a chunk alone overlaps nothing)
    sizes = lane_groups((N + C - 1) // C, NL)
    try:
        ch0 = 0
        for gsz in sizes:
            lanes = [(lane, (ch0 + lane) * C) for lane in range(gsz)]
            ch0 += gsz
            for k, w in enumerate(plan.windows):
                for lane, c0 in lanes:
                    d = w["lane_decs"][lane]
                    chunk, acc = det[c0:c0 + C], pred[c0:c0 + C]
                    B = chunk.shape[0]
                    err = err_l[lane][:B * w["graph"].words].view(B, w["graph"].words)
                    st = st_all[k, c0:c0 + B] if stats is not None else st_all[0, lane * C:lane * C + B]
                    upd = upd_l[lane][:B] if k > 0 else None
                    if post_done[lane] is not None:
                        s_bp.wait_event(post_done[lane])
                    if chain is not None and k == 0:
                        s_bp.wait_event(chain["ready"][c0 // C])   # (the post stream follows through bp_done)
                    d.decode(chunk, w["row0"], upd, err_bits=err, status=st, stage=1, stream=s_bp)
                    bp_done = torch.cuda.Event()
                    bp_done.record(s_bp)
                    d.post_head_start(s_bp)            # (heavy post-processing gets onto the CUs before the next BP kernel fills them; decided on the device)
                    s_post.wait_event(bp_done)
                    d.decode(chunk, w["row0"], upd, err_bits=err, status=st, stage=2, stream=s_post)
                    w["L"].xor_apply(err, acc, accumulate=True, stream=s_post)
                    if w["U"] is not None:
                        w["U"].xor_apply(err, upd_l[lane][:B], accumulate=False, stream=s_post)
                    e = torch.cuda.Event()
                    e.record(s_post)
                    post_done[lane] = e
                    if stats is not None:
                        stats.append((k, st))
    except BaseException:
        # the buffers above were allocated on the caller's stream and are in use on the side streams: nothing may be handed back
        # to the allocator while queued kernels still write to them
        s_bp.synchronize()
        s_post.synchronize()
        raise
    if chain is not None:
        chain["post_done"] = post_done
        chain["s_post"] = s_post
        chain["s_bp"] = s_bp
        return pred                        # (not joined: the caller queues its copy on the post stream and synchronises the side streams at the end)
    for e in post_done:
        if e is not None:
            cur.wait_event(e)
    return pred


def _kwargs_for_device(d, cls):
    """Keyword arguments of plug-in class `cls` -> BatchDecoder options.  The post-processor follows the CLASS, as it does in
    ldpc: a BpLsdDecoder runs LSD whether or not the dict names `lsd_method` / `lsd_order` (ldpc's defaults 'lsd_0', 0).
    Keywords that do not change the algorithm here are dropped (`omp_thread_count`; `input_vector_type` 'syndrome' / 'auto';
    `random_schedule_seed` 0 / None and `serial_schedule_order` None = the natural serial order, which is what this build
    runs); legal ldpc keywords the device path does not implement raise NotImplementedError; the other class's post-processor
    options and unknown names raise TypeError naming the device path (ldpc's own classes take **kwargs and may ignore them:
    refusing is the safe side of "never a silent change of algorithm")."""
    from .bplsd import BpLsdDecoder, lsd_to_device_method
    common = ("bp_method", "schedule", "max_iter", "ms_scaling_factor")
    rates = ("error_rate", "channel_probs", "error_channel")
    d = dict(d)
    is_lsd = isinstance(cls, type) and issubclass(cls, BpLsdDecoder)
    own = ("lsd_method", "lsd_order", "bits_per_step") if is_lsd else ("osd_method", "osd_order")
    d.pop("omp_thread_count", None)
    if str(d.pop("input_vector_type", "syndrome")).lower() not in ("syndrome", "auto"):
        raise NotImplementedError("the device path decodes syndromes only (input_vector_type='syndrome')")
    if d.pop("random_schedule_seed", 0) not in (0, None) or d.pop("serial_schedule_order", None) is not None:
        raise NotImplementedError("the device path runs the serial schedule in natural fault order only (ldpc's default: "
                                  "random_schedule_seed=0, serial_schedule_order=None)")
    extra = [k for k in d if k not in common + rates + own]
    if extra:
        raise TypeError("%s on the device path does not take the keyword argument(s): %s" % (cls.__name__, ", ".join(sorted(extra))))
    out = {k: d[k] for k in d if k in common}
    if is_lsd:
        out["osd_method"], out["osd_order"] = lsd_to_device_method(d.get("lsd_method", "lsd_0"), d.get("lsd_order", 0),
                                                                   d.get("bits_per_step", 1))
    else:
        out.update({k: d[k] for k in d if k in own})
    return out


# ---- plan cache ---------------------------------------------------------------------------------------------------------------
# The reference entry points are called once per experiment point with host arrays (bposd.py:54-86; doc/06B_end_to_end_demo_bb.ipynb
# cell 5 loops over p), and every call used to rebuild everything: DEM extraction + spacetime() + one qd_graph_create per window
# (0.4 s for the headline window, 15 s for the QLP [[1020,136]] circuit).  Plans are kept, least recently used first out, keyed on
# everything they depend on: the circuit (hash of its text), hz, W, F, the number of rounds, both plug-in classes and both option
# dicts -- and the CUDA device that is current, because graphs, decoders and workspaces are bound to the device they were built on.
# QD_PLAN_CACHE = number of plans kept (default 8, 0 = off).  ONE cache per process (the per-thread caches of round 5 rebuilt
# the plan and kept a second set of graphs, decoders and workspaces per calling thread), guarded by a module lock; a plan carries mutable
# state (staging buffers, side streams, decoder workspaces), so its USE is serialised by the plan's own re-entrant lock
# (DeviceWindowPlan.decode_host): two threads calling with the same arguments share one plan and take turns, threads with different
# arguments run different plans side by side.  Only plans that are in use keep device workspaces: a lookup releases the workspaces of
# every other cached plan whose lock is free (whatever thread used it last); a plan that another thread is decoding with is left alone.
import threading as _threading
from collections import OrderedDict as _OrderedDict

_CACHE = _OrderedDict()
_CACHE_LOCK = _threading.RLock()
_CACHE_STATS = {"hits": 0, "misses": 0}


def _freeze(v):
    """Hashable fingerprint of an option value: arrays by content, floats by repr."""
    import hashlib
    if isinstance(v, dict):
        return tuple(sorted((str(k), _freeze(x)) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    if isinstance(v, np.ndarray) or hasattr(v, "__array__"):
        a = np.ascontiguousarray(np.asarray(v))
        return ("ndarray", a.dtype.str, a.shape, hashlib.sha1(a.tobytes()).hexdigest())
    if isinstance(v, float):
        return ("float", repr(float(v)))
    return (type(v).__name__, repr(v))


def _circuit_fingerprint(circuit):
    """sha1 of the circuit text (stim.Circuit, quits_amd.dem.Circuit and plain text all print as Stim text); an object that is
    already a detector error model goes by its printed form as well."""
    import hashlib
    return hashlib.sha1(str(circuit).encode()).hexdigest()


def _matrix_fingerprint(mat):
    import hashlib
    if hasattr(mat, "tocsr"):
        c = mat.tocsr()
        c.sort_indices()
        h = hashlib.sha1(np.asarray(c.indptr, np.int64).tobytes())
        h.update(np.asarray(c.indices, np.int64).tobytes())
        h.update((np.asarray(c.data) % 2).astype(np.uint8).tobytes())
        return ("sparse", c.shape, h.hexdigest())
    a = np.ascontiguousarray(np.asarray(mat) % 2).astype(np.uint8)
    return ("dense", a.shape, hashlib.sha1(a.tobytes()).hexdigest())


def plan_key(kind, circuit, hz, lz, W, F, num_rounds, decoder1, decoder2, dict1, dict2):
    """Everything a DeviceWindowPlan depends on.  kind: 'circuit' (circuit is the circuit) or 'phenom' (circuit is None; lz enters
    because the phenomenological commit matrices are built from it)."""
    return (kind, ("device", _current_device()), None if circuit is None else _circuit_fingerprint(circuit), _matrix_fingerprint(hz),
            None if lz is None else _matrix_fingerprint(lz), int(W), int(F), int(num_rounds),
            getattr(decoder1, "__qualname__", repr(decoder1)), getattr(decoder2, "__qualname__", repr(decoder2)),
            _freeze(dict1), _freeze(dict2))


def _plan_cache_size():
    import os
    try:
        return max(0, int(os.environ.get("QD_PLAN_CACHE", "8")))
    except ValueError:
        return 8


def cached_plan(key, build):
    """The plan stored under `key` in the process-wide cache, built with build() on a miss (under the cache lock: two threads asking for
    the same new plan build it once)."""
    cap = _plan_cache_size()
    with _CACHE_LOCK:
        if cap == 0:
            _CACHE_STATS["misses"] += 1
            return build()
        plan = _CACHE.get(key)
        # only plans in use keep device workspaces (the per-edge BP kernel sizes its message planes for tens of GB): the others keep
        # their graphs and decoders and size their workspaces again when they are used next
        for k, other in _CACHE.items():
            if k != key and hasattr(other, "release_workspaces") and getattr(other, "_ws_live", True):
                lock = getattr(other, "_lock", None)
                if lock is None:
                    other.release_workspaces()
                    other._ws_live = False
                elif lock.acquire(blocking=False):           # (busy in another thread: leave it)
                    try:
                        other.release_workspaces()
                        other._ws_live = False
                    finally:
                        lock.release()
        if plan is not None:
            _CACHE.move_to_end(key)
            _CACHE_STATS["hits"] += 1
        else:
            _CACHE_STATS["misses"] += 1
            plan = build()
            _CACHE[key] = plan
            while len(_CACHE) > cap:
                _CACHE.popitem(last=False)
        try:
            plan._ws_live = True
        except AttributeError:
            pass
        return plan


def plan_cache_info():
    with _CACHE_LOCK:
        return {"size": len(_CACHE), "capacity": _plan_cache_size(), **_CACHE_STATS}


def plan_cache_clear():
    with _CACHE_LOCK:
        _CACHE.clear()
        _CACHE_STATS.update(hits=0, misses=0)


def build_circuit_plan(circuit, hz, W, F, num_rounds, dict1, dict2, decoder1=None, decoder2=None):
    from .bposd import BpOsdDecoder
    nz = hz.shape[0]
    num_cor_rounds, _, _ = window_count(num_rounds, W, F)
    checks, commits, priors, updates = spacetime(circuit, hz, W, F, num_cor_rounds)
    row0 = [F * k * nz for k in range(num_cor_rounds)] + [F * num_cor_rounds * nz]
    return DeviceWindowPlan(checks, commits, priors, updates, row0, nz, commits[0].shape[0],
                            _kwargs_for_device(dict1, decoder1 or BpOsdDecoder), _kwargs_for_device(dict2, decoder2 or BpOsdDecoder))


def phenom_window_set(hz, lz, W, F, num_rounds, rate_mid, rate_last):
    """The phenomenological variant's windows in spacetime()'s format (checks, commits, priors, updates): the analytic window
    matrices of reference sliding_window.py:56-68 with the slicing of :86,:88,:96,:99 written as matrices
    (commit = lz @ sum of the first F data blocks, by linearity; hand-off = measurement block F-1 of the decoded vector)."""
    hz = np.asarray(hz) % 2
    lz = np.asarray(lz) % 2
    nz, nq = hz.shape
    num_cor_rounds, W_last, _ = window_count(num_rounds, W, F)
    h_mid, h_last = phenom_window_matrices(hz, W, F, W_last)
    commit_mid = csr_matrix(np.concatenate([np.tile(lz, (1, F)), np.zeros((lz.shape[0], h_mid.shape[1] - F * nq), int)], axis=1))
    commit_last = csr_matrix(np.concatenate([np.tile(lz, (1, W_last)), np.zeros((lz.shape[0], h_last.shape[1] - W_last * nq), int)], axis=1))
    sel = np.zeros((nz, h_mid.shape[1]), dtype=int)
    sel[np.arange(nz), W * nq + (F - 1) * nz + np.arange(nz)] = 1
    checks = [h_mid] * num_cor_rounds + [h_last]
    commits = [commit_mid] * num_cor_rounds + [commit_last]
    updates = [csr_matrix(sel)] * num_cor_rounds
    priors = [np.full(h_mid.shape[1], float(rate_mid))] * num_cor_rounds + [np.full(h_last.shape[1], float(rate_last))]
    return checks, commits, priors, updates


def build_phenom_plan(hz, lz, W, F, num_rounds, dict1, dict2, decoder1=None, decoder2=None):
    from .bposd import BpOsdDecoder
    nz = np.asarray(hz).shape[0]
    num_cor_rounds, _, _ = window_count(num_rounds, W, F)
    checks, commits, priors, updates = phenom_window_set(hz, lz, W, F, num_rounds, dict1["error_rate"], dict2["error_rate"])
    row0 = [F * k * nz for k in range(num_cor_rounds)] + [F * num_cor_rounds * nz]
    return DeviceWindowPlan(checks, commits, priors, updates, row0, nz, np.asarray(lz).shape[0],
                            _kwargs_for_device(dict1, decoder1 or BpOsdDecoder), _kwargs_for_device(dict2, decoder2 or BpOsdDecoder))


def sliding_window_phenom_mem(zcheck_samples, hz, lz, W, F, decoder1, decoder2, dict1: dict, dict2: dict,
                              function_name1: str, function_name2: str, tqdm_on=False):
    """Phenomenological sliding-window decoder with plug-in inner decoders (reference sliding_window.py:14-101).

    :return logical_z_pred: int64 array (# trials, # logical qubits)
    """
    if F == 0:
        raise ValueError("Input parameter F cannot be zero.")
    nz, nq = hz.shape
    num_trials = zcheck_samples.shape[0]
    num_rounds = zcheck_samples.shape[1] // nz - 2
    num_cor_rounds, W_last, whole = window_count(num_rounds, W, F)
    if whole:
        warnings.warn("Window size larger than the syndrome extraction rounds: Doing whole history correction")

    if _is_device_decoder(decoder1) and _is_device_decoder(decoder2) and function_name1 == function_name2 == "decode":
        plan = cached_plan(plan_key("phenom", None, hz, lz, W, F, num_rounds, decoder1, decoder2, dict1, dict2),
                           lambda: build_phenom_plan(hz, lz, W, F, num_rounds, dict1, dict2, decoder1, decoder2))
        return plan.decode_host(zcheck_samples)

    h_mid, h_last = phenom_window_matrices(hz, W, F, W_last)
    dec_mid = decoder1(h_mid, **dict1)
    dec_last = decoder2(h_last, **dict2)
    samples = np.asarray(zcheck_samples)
    out = np.zeros((num_trials, lz.shape[0]), dtype=int)
    for i in _progress(range(num_trials), tqdm_on):
        total = np.zeros(nq, dtype=int)
        carry = np.zeros(nz, dtype=int)
        for k in range(num_cor_rounds):
            s = samples[i, F * k * nz:(F * k + W) * nz].copy() % 2
            s[:nz] = (s[:nz] + carry) % 2
            e = getattr(dec_mid, function_name1)(s)
            total = (total + np.sum(e[:F * nq].reshape(F, nq), axis=0)) % 2
            carry = e[W * nq + (F - 1) * nz:W * nq + F * nz].copy()
        s = samples[i, F * num_cor_rounds * nz:].copy() % 2
        s[:nz] = (s[:nz] + carry) % 2
        e = getattr(dec_last, function_name2)(s)
        total = (total + np.sum(e[:W_last * nq].reshape(W_last, nq), axis=0)) % 2
        out[i, :] = (lz @ total) % 2
    return out


def sliding_window_circuit_mem(zcheck_samples, circuit, hz, lz, W, F, decoder1, decoder2, dict1: dict, dict2: dict,
                               error_rate_name1: str, error_rate_name2: str,
                               function_name1: str, function_name2: str, tqdm_on=False):
    """Circuit-level (space-time detector error model) sliding-window decoder with plug-in inner decoders
    (reference sliding_window.py:104-188).  `circuit`: a stim.Circuit, circuit text, or quits_amd.dem.Circuit.

    :return logical_z_pred: int64 array (# trials, # logical qubits)
    """
    if F == 0:
        # the reference only reaches this message through spacetime() (base.py:149-150) and, when R + 2 >= W, dies
        # earlier on the integer division at sliding_window.py:135; one ValueError up front covers both
        raise ValueError("Input parameter F cannot be zero.")
    nz = hz.shape[0]
    num_trials = zcheck_samples.shape[0]
    num_rounds = zcheck_samples.shape[1] // nz - 2
    num_cor_rounds, _, whole = window_count(num_rounds, W, F)
    if whole:
        warnings.warn("Window size larger than the syndrome extraction rounds: Doing whole history correction")

    if _is_device_decoder(decoder1) and _is_device_decoder(decoder2) and function_name1 == function_name2 == "decode":
        plan = cached_plan(plan_key("circuit", circuit, hz, None, W, F, num_rounds, decoder1, decoder2, dict1, dict2),
                           lambda: build_circuit_plan(circuit, hz, W, F, num_rounds, dict1, dict2, decoder1, decoder2))
        return plan.decode_host(zcheck_samples)

    checks, commits, priors, updates = spacetime(circuit, hz, W, F, num_cor_rounds)
    decoders = []
    for k in range(len(checks)):
        last = k == len(checks) - 1
        kw = dict(dict2 if last else dict1)          # the reference writes into the caller's dicts (:148,:151); we don't
        kw[error_rate_name2 if last else error_rate_name1] = priors[k]
        decoders.append((decoder2 if last else decoder1)(checks[k], **kw))
    samples = np.asarray(zcheck_samples)
    out = np.zeros((num_trials, lz.shape[0]), dtype=int)
    for i in _progress(range(num_trials), tqdm_on):
        acc = np.zeros(commits[0].shape[0], dtype=int)
        carry = np.zeros(nz, dtype=int)
        for k in range(num_cor_rounds):
            s = samples[i, F * k * nz:(F * k + W) * nz].copy() % 2
            s[:nz] = (s[:nz] + carry) % 2
            e = getattr(decoders[k], function_name1)(s)
            ncommit = commits[k].shape[1]
            acc = (acc + commits[k] @ e[:ncommit] % 2) % 2
            carry = updates[k] @ e[:ncommit] % 2
        s = samples[i, F * num_cor_rounds * nz:].copy() % 2
        s[:nz] = (s[:nz] + carry) % 2
        e = getattr(decoders[num_cor_rounds], function_name2)(s)
        acc = (acc + commits[num_cor_rounds] @ e % 2) % 2
        out[i, :] = acc
    return out


__all__ = ["sliding_window_phenom_mem", "sliding_window_circuit_mem", "plan_cache_info", "plan_cache_clear"]
