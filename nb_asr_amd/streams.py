"""Streams that really run side by side (round 6).

A HIP stream is placed on one of a few hardware queues per priority when it is created (four by default on this runtime; the stream
pool of PyTorch hands out 32 streams per priority over them: on the GPU box pool streams 0-7 of normal priority sat on queues
a b c c b a d c, the default stream on d).  Two streams on ONE queue execute their kernels one after the other -- a pipelined
forward's tail then no longer overlaps the next encoder, and with two chains of forwards in flight (`ASRModel.forward_many`) the rate
falls to a fraction (`tools/ubench/stream_pairs.py`, `collapse_repro.py`: 7 100 -> 900 utterances/s at 8 utterances once re-created
plans had drawn the "wrong" pool streams).  A high-priority stream brings a second hazard: its queue shares a pipe with one of the
normal-priority queues and its 250 short kernels per forward starve whatever encoder runs there.  Priority buys the tail nothing
measurable (10 528 vs 10 548 utterances/s at 64 x 1000), so the tails run at normal priority and every stream this package creates is
CHOSEN: candidates (streams of our own, hipStreamCreateWithFlags) are probed with one-thread spin kernels against the streams they must overlap with.

The tail stream of a main stream is chosen once and remembered (a plan that is dropped and built again gets the same one), and
`forward_many` chooses its chain streams AND their tail streams before it starts its threads: the probe waits for the device, which
must not happen while another thread captures a graph (the cached recurrence chain) -- HIP invalidates that capture.
"""
import ctypes
import threading
import time

import torch

_lock = threading.RLock()
_used = {}                   # device index -> [(kind, stream)]: 'chain' (forward_many's streams), 'chain_tail', 'tail'
_tails = {}                  # (device index, main stream handle) -> its tail stream
_spare = {}                  # device index -> candidates that were created, probed and not taken (tried first by the next choice)
_verdicts = {}               # (device index, handle a, handle b) -> bool
_single = {}                 # device index -> seconds of one spin kernel
SPIN_CYCLES = 150_000
MAX_CANDIDATES = 8
WEIGHT = {'chain': 100, 'chain_tail': 10, 'tail': 1}


def _new_stream(device):
    """A stream of our own (nbasr_stream_create: hipStreamCreateWithFlags, non-blocking), wrapped for torch: PyTorch's pool has 32
    streams per priority and hands them out round-robin -- a dozen probes later a "new" pool stream IS one that is already in use (by
    us or by the caller)."""
    from . import hip
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        hip._check(hip.load_library().nbasr_stream_create(ctypes.byref(handle)), 'nbasr_stream_create')
    return torch.cuda.ExternalStream(handle.value, device=device)


def _run(device, streams, reps=3, whole_device=True):
    best = None
    for _ in range(reps):
        if whole_device:
            torch.cuda.synchronize(device)
        else:
            for s in streams:
                s.synchronize()
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(SPIN_CYCLES)
        for s in streams:
            s.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def overlaps(device, a, b, whole_device=True):
    """Do kernels of streams ``a`` and ``b`` execute at the same time?  (False for a stream with itself.)"""
    device = torch.device(device)
    if a.cuda_stream == b.cuda_stream:
        return False
    key = (device.index,) + tuple(sorted((a.cuda_stream, b.cuda_stream)))
    with _lock:
        if key not in _verdicts:
            if device.index not in _single:
                _run(device, [a], reps=2, whole_device=whole_device)          # clocks up
                _single[device.index] = _run(device, [a], whole_device=whole_device)
            _verdicts[key] = _run(device, [a, b], whole_device=whole_device) < 1.5 * _single[device.index]
        return _verdicts[key]


def _pick(device, must, kind, whole_device=True):
    """A normal-priority stream of our own that overlaps with every stream in ``must`` and with as much of what is in use as possible
    (weighted: a chain's stream is always busy while `forward_many` runs, its tail stream nearly so, a lone plan's tail stream only
    between calls); registered as ``kind``.  Candidates: the ones earlier choices created and did not take, then up to MAX_CANDIDATES new
    ones.  Falls back to the least conflicting candidate (never fails)."""
    used = _used.setdefault(device.index, [])
    spare = _spare.setdefault(device.index, [])
    best, best_score, tried = None, None, []
    taken = {s.cuda_stream for _, s in used} | {m.cuda_stream for m in must}
    created = 0
    while True:
        if len(tried) < len(spare):
            cand = spare[len(tried)]
        elif created < MAX_CANDIDATES:
            cand, created = _new_stream(device), created + 1
            spare.append(cand)
        else:
            break
        tried.append(cand)
        if cand.cuda_stream in taken:
            continue
        score = sum(1000 for s in must if not overlaps(device, cand, s, whole_device))
        for k, s in used:
            if any(s.cuda_stream == m.cuda_stream for m in must):
                continue
            # beside other threads only pairs that were probed earlier count: a spin kernel launched into a stream another thread is
            # capturing (the cached recurrence graph) would invalidate that capture
            known = (device.index,) + tuple(sorted((cand.cuda_stream, s.cuda_stream))) in _verdicts
            if (whole_device or known) and not overlaps(device, cand, s, whole_device):
                score += WEIGHT[k]
        if best is None or score < best_score:
            best, best_score = cand, score
        if score == 0:
            break
    if best is None:
        best = _new_stream(device)
    spare[:] = [c for c in spare if c is not best]
    used.append((kind, best))
    return best


def tail_stream_for(device, main, whole_device=True):
    """The stream a pipelined forward on ``main`` runs its LSTM + head on: chosen at the first request, then the same one every time.
    ``whole_device=False``: the probe waits for the probed streams only (a caller that cannot rule out a capture in another thread)."""
    device = torch.device(device)
    key = (device.index, main.cuda_stream)
    with _lock, torch.cuda.device(device):
        if key not in _tails:
            is_chain = any(k == 'chain' and s.cuda_stream == main.cuda_stream for k, s in _used.get(device.index, ()))
            _tails[key] = _pick(device, [main], 'chain_tail' if is_chain else 'tail', whole_device)
        return _tails[key]


def chain_streams(device, ways, have=()):
    """``ways`` more streams for chains of forwards in flight (`ASRModel.forward_many`), pairwise (and with ``have``) on different
    hardware queues, each with its tail stream chosen as well.  Call before the chains' threads start."""
    device = torch.device(device)
    out = list(have)
    with _lock, torch.cuda.device(device):
        for _ in range(ways):
            out.append(_pick(device, out, 'chain'))
        for s in out:
            tail_stream_for(device, s)
    return out[len(have):]
