"""Host-side executor: walks the model's layer list and enqueues the HIP kernels.

Replaces the isinstance-dispatch loop of the reference's ``ASRModel.forward`` (model.py:116-131),
``SearchCell.forward`` (model.py:49-59) and ``Node.forward`` (model.py:13-22):

* activations stay in (batch, channels, frames) order in pitched workspace buffers (row pitch =
  frames rounded up to 4, pitch columns kept at zero), so none of the reference's permutes, padded
  copies, ``zeros_like`` branches or ``0 + x`` adds exist here;
* a node is ONE launch: main op + bias + ReLU + clamp + the sum of its flagged skip inputs;
* LayerNorm runs on the channel (slow) dimension in place.

Four workspace buffers sized for the widest block are rotated; nothing is allocated per call
except the returned logits.  A plan belongs to a DEVICE, not to a (batch, frames) shape and not to a model
object: its workspaces only ever grow (a stream of TIMIT batches whose length differs every step re-uses the
same memory), and the model is passed to every call, so ``torch.nn.DataParallel`` replicas (shallow copies of
the module that share the plan pool) each run with their own parameters.
"""
import collections
import math
import os
import threading
import weakref

import torch

from . import hip


def node_into(node, inputs, frames, out, ln0=None, stats=None, linear_ctx=None, gc_variant=0):
    """Enqueue one cell node: ``out = op(inputs[-1]) + sum(flagged inputs)`` (left-to-right).

    ``ln0`` = (stats, gamma, beta): ``inputs[0]`` (the cell input) is stored un-normalised with a pending LayerNorm that
    the kernel applies while loading -- as the main input when the node is the cell's first, as skip0 when flagged.
    ``stats`` = (stats_out, workspace, eps): a grouped-conv node also emits the LayerNorm statistics of ``out``.
    ``linear_ctx`` = (packed_weights(op), workspace(c_in, ld)) callables of a ForwardPlan: `linear` ops then run on the fp16
    matrix cores (packed weights, pre-split activations); None = the exact-fp32 MFMA GEMM."""
    from .ops import PadConvRelu, Linear, Zero, Identity
    if len(inputs) != len(node.branch_ops):
        raise AssertionError('Branch op and input list have different lenghts')
    skips = [src for branch, src in zip(node.branch_ops, inputs) if isinstance(branch, Identity)]
    op, last = node.op, inputs[-1]
    on_x = ln0 is not None and len(inputs) == 1
    on_s0 = ln0 is not None and isinstance(node.branch_ops[0], Identity)       # the cell input, if flagged, is skips[0]
    ln = ln0 if (on_x or on_s0) else None
    if isinstance(op, PadConvRelu):
        # with `stats` the epilogue writes partial statistics to the workspace; the caller merges them (finalize)
        hip.grouped_conv1d_node(last, op.conv.weight.detach(), op.conv.bias.detach(), skips, out, frames,
                                op.groups, op.kernel_size, op.dilation, ln, on_x, on_s0,
                                stats[1] if stats is not None else None, gc_variant)
    elif isinstance(op, Linear):
        if stats is not None:
            raise ValueError('statistics from the epilogue are only available for grouped-conv nodes')
        if linear_ctx is not None:
            packed, workspace = linear_ctx
            hip.linear_fused_packed(last, frames, packed(op.linear), op.linear.out_features, op.linear.bias.detach(), skips, out,
                                    workspace(last.shape[1], last.shape[2]), ln, on_x, on_s0)
        else:
            hip.dense_conv1d_fused(last, frames, op.linear.weight.detach().unsqueeze(-1), op.linear.bias.detach(),
                                   skips, out, 1, ln, on_x, on_s0)
    elif isinstance(op, Zero):
        hip.skip_sum(skips, out, frames, ln if on_s0 else None, on_s0)
    else:
        raise TypeError(f'unsupported node operation {type(op).__name__}')
    return out


class PendingLogits:
    """Logits produced on the plan's side stream (pipelined forward).  ``result()`` makes the caller's current stream
    wait for them -- until then the next batch's encoder may already be running on the main stream."""

    def __init__(self, logits, event):
        self._logits, self._event = logits, event
        self._seq = None             # (plan, event behind the status copy, ring slot) when this forward ran the one-launch recurrence

    def result(self):
        """The logits, ordered behind the tail on the caller's current stream (no host wait) -- except for a forward whose LSTM
        recurrence ran as ONE resident launch (NBASR_LSTM_SEQ=1 in pipelined mode): that launch can time out when another process
        takes its compute units, so ``result()`` then waits for ITS status word and raises ``HipError`` instead of handing out the
        NaN rows.  A plain ``model(x)`` returns a tensor, not a handle: it can only report at the next call or
        through ``model.check()``."""
        if self._event is not None:
            cur = torch.cuda.current_stream(self._logits.device)
            cur.wait_event(self._event)
            self._logits.record_stream(cur)
            self._event = None
        if self._seq is not None:
            (plan, ev, slot), self._seq = self._seq, None
            plan.check_seq_slot(ev, slot)
        return self._logits


class GroupLogits(PendingLogits):
    """Handle of a pipelined forward whose LSTM + head run together with those of the other forwards of its TAIL GROUP (round 6,
    `ASRModel.forward_many(tail_group=...)`): the logits exist once the group's last forward has been enqueued."""

    def __init__(self):
        super().__init__(None, None)

    def _set(self, logits, event):
        self._logits, self._event = logits, event

    def result(self):
        if self._logits is None:
            raise hip.HipError('this forward belongs to a tail group whose last forward has not been enqueued yet '
                               '(forward_many enqueues a whole group before it reads a result)')
        return super().result()


def _load_gc_table():
    import json
    import pathlib
    path = pathlib.Path(__file__).with_name('gc_variant_table.json')
    return json.loads(path.read_text()) if path.exists() else {}


_GC_TABLE = _load_gc_table()


def _load_dense_tile_table():
    """Measured us per launch of the image-path dense convolution per (row tile, frame tile): nb_asr_amd/dense_tile_table.json
    (tools/ubench/dense_tiles.py -> tools/make_dense_tile_table.py), {(c_in, c_out, stride, frames_out): {batch: {(rows, frames): us}}}."""
    import json
    import pathlib
    path = pathlib.Path(__file__).with_name('dense_tile_table.json')
    if not path.exists():
        return {}
    raw = json.loads(path.read_text())['table']
    return {tuple(int(v) for v in k.split(',')): {int(b): {tuple(int(v) for v in t.split('x')): us for t, us in row.items()} for b, row in by_b.items()}
            for k, by_b in raw.items()}


_DENSE_TILES = _load_dense_tile_table()


# Any submodule or parameter (re-)registered on any module bumps this counter: a launch tape recorded before is not replayed
# afterwards (a layer swapped inside a model that has already run must be seen by the next forward).
_structure_epoch = [0]


def _bump_structure_epoch(*_args):
    _structure_epoch[0] += 1


torch.nn.modules.module.register_module_module_registration_hook(_bump_structure_epoch)
torch.nn.modules.module.register_module_parameter_registration_hook(_bump_structure_epoch)


class LaunchTape:
    """The enqueueing C-ABI calls of ONE forward, recorded once and replayed for every later batch with the same key
    (shape, dtype, streams, parameter addresses and versions): the plan's workspaces and the derived weight copies sit
    at fixed addresses, so the only arguments that change are the caller's input pointer and the freshly allocated
    logits, which are patched into the recorded argument lists.  Host-side steps that belong to the sequence (event
    record / wait between the main and the side stream, the logits allocation) are recorded as callables.

    Why: a forward is ~380 launches, ~130 of them issued from python at ~12 us each; at 8 utterances per GPU (the
    strong-scaling split of the benchmark batch) that host time, 2.4 ms, exceeds the device time.  A replay issues the same
    calls in ~0.4 ms (tools/ubench/host_issue.py)."""

    def __init__(self, entries, x_ptr, pipelined, pipe, group=None):
        self.entries, self.pipelined, self.pipe, self.group = entries, pipelined, pipe, group
        self.x_slots = []
        producers = [(i, e) for i, e in enumerate(entries) if e[0] is None and e[2] is not None]
        for i, e in enumerate(entries):
            if e[0] is None:
                continue
            for j, a in enumerate(e[1]):
                if type(a) is not int:
                    continue
                if a == x_ptr:
                    self.x_slots.append((i, j))
                for pi, pe in producers:
                    if pi < i and a == pe[2]:
                        pe[3].append((i, j))
        self.launches = sum(1 for e in entries if e[0] is not None)

    def replay(self, plan, x):
        entries = self.entries
        if not self.pipe:
            plan.wait_tails()
        px = x.data_ptr()
        for i, j in self.x_slots:
            entries[i][1][j] = px
        for e in entries:
            fn = e[0]
            if fn is None:
                t = e[1]()
                if e[3]:
                    p = t.data_ptr()
                    for i, j in e[3]:
                        entries[i][1][j] = p
            else:
                rc = fn(*e[1])
                if rc:
                    hip._check(rc, fn.__name__)
        logits, plan._tape_logits = plan._tape_logits, None
        if self.group is not None:
            return plan._group_member(self.group, logits)
        if self.pipelined:
            return PendingLogits(logits, plan.tail_done[plan._turn] if self.pipe else None)
        return logits


# Environment switches of earlier rounds that no longer exist: a script that still sets one gets ONE warning instead of silence
_REMOVED_SWITCHES = {
    'NBASR_TRAIN_GEMM': 'folded into NBASR_DENSE_MODE (f32 = every GEMM on the exact-fp32 MFMA)',
    'NBASR_IMAGE_MODE': None, 'NBASR_ROW_TILE': None, 'NBASR_LN_MODE': None, 'NBASR_EPILOGUE_STATS': None, 'NBASR_LSTM_UNPACKED': None,
    'NBASR_GC_TABLE': 'NBASR_GC_F32_VARIANT=0 runs the default node kernel everywhere', 'NBASR_GC_BF16_VARIANT': None,
    'NBASR_GC_BF16_MFMA': 'NBASR_CELL_FUSION=1|valu|0 chooses the bf16 cell kernel', 'NBASR_MFMA_SPLITS': None,
    'NBASR_DENSE_EPILOGUE': None, 'NBASR_FORCE_DIST': None, 'NBASR_PW_ORDER': None,
}
_warned_removed = [False]


def _warn_removed_switches():
    if _warned_removed[0]:
        return
    _warned_removed[0] = True
    stale = [k for k in _REMOVED_SWITCHES if k in os.environ]
    if stale:
        import warnings
        notes = '; '.join(f'{k} ({_REMOVED_SWITCHES[k]})' if _REMOVED_SWITCHES[k] else k for k in stale)
        warnings.warn(f'nb_asr_amd: these environment switches were removed and are IGNORED: {notes}.  The current ones: NBASR_DENSE_MODE, '
                      f'NBASR_LINEAR_MODE, NBASR_CELL_FUSION, NBASR_GC_F32_VARIANT, NBASR_LSTM_SEQ, NBASR_TAPE, NBASR_CONV_STATS (DESIGN.md 5)', stacklevel=3)


class ForwardPlan:
    """Workspaces + launch sequence for one device.  Shape-independent: every buffer is flat, grows on demand
    (never shrinks) and is viewed per call; at most one forward is being ENQUEUED through a plan at a time
    (``PlanPool`` hands a plan to one thread at a time), while pipelined tails of earlier calls may still be in flight."""

    def __init__(self, device):
        _warn_removed_switches()
        self.device = torch.device(device)
        self.batch = self.frames = self.out_frames = 0
        self.block_frames = []
        self.timer = None
        # dense k=8 convs, all fp32-accurate:
        #   'auto' (default) = 2-way fp16 split (3 MFMAs per product) wherever the input has just been written by the
        #                      LayerNorm kernel (which also emits the per-utterance max|x| the scheme's range scaling
        #                      needs), the 3-way bf16 split (6 MFMAs per product, fp32's exponent range) elsewhere;
        #   'bf16x3' = 3-way bf16 split everywhere;  'f32' = the exact-fp32 MFMA kernel
        self.dense_mode = os.environ.get('NBASR_DENSE_MODE', 'auto')
        if self.dense_mode not in ('auto', 'bf16x3', 'f32'):
            raise ValueError(f'NBASR_DENSE_MODE must be auto, bf16x3 or f32, got {self.dense_mode!r}')
        self._packed = {}            # (id(parameter), tag) -> (weakref(parameter), version, packed tensor)
        # per-frame linear maps (`linear` node ops, LSTM input projection): 'f16x2' = fp16 matrix cores with pre-split
        # activations (default), 'f32' = the exact-fp32 MFMA GEMM
        self.linear_mode = os.environ.get('NBASR_LINEAR_MODE', 'f16x2')
        if self.linear_mode not in ('f16x2', 'f32'):
            raise ValueError(f'NBASR_LINEAR_MODE must be f16x2 or f32, got {self.linear_mode!r}')
        # (round 3: the A/B switches of rounds 1-2 whose alternatives lost everywhere are gone -- NBASR_IMAGE_MODE, NBASR_ROW_TILE,
        # NBASR_LN_MODE, NBASR_EPILOGUE_STATS, NBASR_LSTM_UNPACKED, NBASR_GC_TABLE, NBASR_GC_BF16_VARIANT, NBASR_GC_BF16_MFMA; what is
        # left: NBASR_DENSE_MODE (also read by autograd.py), NBASR_LINEAR_MODE, NBASR_CELL_FUSION, NBASR_GC_F32_VARIANT, NBASR_LSTM_SEQ, NBASR_TAPE, NBASR_CONV_STATS)
        self._act_image = None
        self.dense_schemes = {}      # block -> scheme used by the last run (read by bench.py)
        self.dense_row_tiles = {}    # block -> rows per workgroup of the image-path GEMM in the last run
        self.dense_frame_tiles = {}  # block -> frames per workgroup (256, or 128 where the measured table says so)
        # cells whose three nodes are grouped convs run as ONE launch where a row fits a workgroup (<= 1024 frames): x1 and x2 never
        # touch HBM and the cell's LayerNorm statistics come out of the same launch (grouped_cell.hip, round 3).  Bit-identical to
        # the three node launches; NBASR_CELL_FUSION=0 turns it off (A/B)
        _cf = os.environ.get('NBASR_CELL_FUSION', '1')
        if _cf not in ('1', '0', 'valu'):
            raise ValueError(f"NBASR_CELL_FUSION={_cf!r}: expected '1', '0' or 'valu'")
        self.cell_fusion = _cf != '0'
        self.cell_mfma = _cf == '1'          # bf16 storage: the matrix-core cell kernel ('valu': the vector-ALU one, as for fp32)
        # the LSTM recurrence (round 6: fp16-pair matrix cores, a tile of 16 utterances per workgroup row): 'auto' = ONE resident launch whose
        # exchange stays inside an XCD (nbasr_lstm_recurrence_xcd) in the plain forward and in a captured graph, the same arithmetic as one
        # launch per frame (nbasr_lstm_recurrence_frames16) in a pipelined tail on the side stream; 'xcd' / 'frames' = that form everywhere
        # ('frames' is also what a failed status word demotes a plan to); the fp32-MFMA kernels of rounds 1-5, another summation order:
        # '0' = one launch per frame, '1' = the chip-wide resident grid (bit-identical to '0') wherever it applies
        self.lstm_seq_mode = os.environ.get('NBASR_LSTM_SEQ', 'auto')
        if self.lstm_seq_mode not in ('auto', 'xcd', 'frames', '0', '1'):
            raise ValueError(f"NBASR_LSTM_SEQ={self.lstm_seq_mode!r}: expected 'auto', 'xcd', 'frames', '0' or '1'")
        self.tail_mode = os.environ.get('NBASR_TAIL', 'auto')
        if self.tail_mode not in ('auto', 'side', 'main'):
            raise ValueError(f"NBASR_TAIL={self.tail_mode!r}: expected 'auto', 'side' or 'main'")
        self._group_handles = []      # handles of the tail group being enqueued (_group_member)
        self._released_on, self._released_event = None, None   # PlanPool: the stream the last forward was enqueued on, an event behind it
        self._seq_host, self._seq_pending = None, None      # pinned ring of status words / deque of (event behind the copy, slot) (check_seq)
        self._seq_slot, self._seq_failed = 0, False
        self._seq_outcome = {}       # ring slot -> whether that launch failed, once its word has been read (check_seq_slot)
        self._seq_lock = threading.Lock()    # PendingLogits.result() may run on another thread than the one enqueueing through the plan
        self._seq_last = None        # (event, slot) of the status copy the forward being enqueued has just added
        # statistics of a downsample convolution's output from its own epilogue (round 5); NBASR_CONV_STATS=0: a pass over the output
        self.conv_stats = os.environ.get('NBASR_CONV_STATS', '1') != '0'
        self._conv_part = None       # partials the dense convolution just enqueued left for the LayerNorm behind it
        self._seq_flags = hip.LSTM_SEQ_INJECT_FAULT if os.environ.get('NBASR_LSTM_SEQ_FAULT') == '1' else 0     # tests: force a timeout
        # fp32 node kernel variant per launch from the measured table (_gc_variant); NBASR_GC_F32_VARIANT=<bits> forces one (0: the
        # default kernel everywhere)
        self.gc_table = _GC_TABLE
        self._bufs = {}              # name -> flat tensor; grow-only (see _buf)
        self.grow_count = 0          # number of (re)allocations so far (tests: a smaller batch must not allocate)
        # pipelined mode (forward_async): the latency-bound LSTM + head of batch i run on a side stream while the main
        # stream already runs the encoder of batch i+1; the encoder output is double-buffered for that
        self.side_stream = None
        self._graphs = {}            # (batch, frames) -> (graph, parameter signature, x_static, y_static)
        self.tail_done = [None, None]
        self._turn = 0
        # launch tapes (LaunchTape): NBASR_TAPE=0 issues every forward through the python launch sequence
        self.tape_mode = os.environ.get('NBASR_TAPE', '1') != '0'
        self._tapes, self._tape_seen = {}, {}
        self._recording = None       # the entries list while a tape is being recorded
        self._mutations = 0          # workspace growths + derived-weight rebuilds (either one invalidates every tape)
        self._tape_logits = None
        self._param_slots = None     # (weakref(model), [(module._parameters, name), ...])
        self.tape_replays = 0

    # ---- grow-only workspaces ---------------------------------------------------------------------------------------
    def _buf(self, name, numel, dtype=torch.float32):
        """Flat workspace ``name`` with at least ``numel`` elements.  Growing re-allocates: pipelined tails that may
        still read the old memory are waited for first (stream-ordered, no host sync), and captured graphs die."""
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            self.wait_tails()
            self._graphs.clear()
            self._tapes.clear()
            self._mutations += 1
            t = self._bufs[name] = torch.empty(max(int(numel), 4), device=self.device, dtype=dtype)
            if self.side_stream is not None:
                t.record_stream(self.side_stream)
            self.grow_count += 1
        return t

    def _recurrence(self, gates, w_hh, hidden, pipe, capturing=False):
        """The LSTM recurrence into ``self.h_out``.  Round 6: ONE resident launch with a tile of 16 utterances per XCD on the fp16 matrix
        cores (nbasr_lstm_recurrence_xcd) wherever it applies -- plain forward, pipelined tail and captured graph alike (every route the same
        arithmetic); a layer wider than 512 or a plan demoted by a failed status word takes one launch per frame (replayed as a graph)."""
        # (launches of the resident grid are chained across the process's streams by an event, which a graph capture cannot hold)
        if not capturing and torch.cuda.is_current_stream_capturing():       # a caller's own torch.cuda.graph(...) around model(x)
            capturing = True
        mode = self.lstm_seq_mode
        # 'auto': the resident launch everywhere except in a tail on the SIDE stream -- a resident grid there fills whole XCDs (32 compute
        # units each, for the length of the recurrence) while the hardware deals every workgroup of the next batch's encoder kernels to a
        # fixed XCD: the encoder stalls on the occupied ones (measured, round 6: 9 620 against 9 995 utterances/s at 64, 4 136 against
        # 5 398 at 8).  There, and in a plan demoted by a failed status word ('frames'), the SAME arithmetic runs as one launch per frame
        # (nbasr_lstm_recurrence_frames16): every route of the default configuration gives the same bits
        if mode in ('auto', 'xcd', 'frames'):
            nbytes = hip.lstm_xcd_workspace_bytes(self.batch, hidden)
            if nbytes:
                ws = self._buf('lstm_xcd', nbytes, torch.uint8)
                if mode == 'xcd' or (mode == 'auto' and not pipe):
                    out = hip.lstm_recurrence_xcd(gates, self._packed_whh16(w_hh), self.cell_ws, self.h_out, ws, self._seq_flags)
                    if not capturing:          # (a captured graph holds the launch, not the host-side read-back of its status word)
                        self._host(lambda: self._seq_status_readback(ws))
                    return out
                return hip.lstm_recurrence_frames16(gates, self._packed_whh16(w_hh), self.cell_ws, self.h_out, ws)
        packed_hh = self._packed_whh(w_hh)
        nbytes = hip.lstm_seq_workspace_bytes(self.batch, hidden, self.device) if (mode == '1' and not capturing) else 0
        if nbytes:
            ws = self._buf('lstm_seq', nbytes, torch.uint8)
            try:
                out = hip.lstm_recurrence_seq(gates, packed_hh, self.cell_ws, self.h_out, ws, self._seq_flags)
            except hip.HipError as e:
                # ONLY the device refusing the cooperative grid (a partition / CU mask smaller than the occupancy query said) demotes the
                # plan to per-frame launches; an argument, alignment or capture error is the caller's to see.  The refusal is
                # recognised by its CODE (hipErrorCooperativeLaunchTooLarge, passed through by the entry point), not by its wording
                if e.code != hip.HIP_ERROR_COOPERATIVE_LAUNCH_TOO_LARGE:
                    raise
                self.lstm_seq_mode = '0'
                self._tapes.clear()
                self._mutations += 1          # a tape being recorded right now holds the refused launch: it must not be kept
                return hip.lstm_recurrence_packed(gates, packed_hh, self.cell_ws, self.h_out)
            self._host(lambda: self._seq_status_readback(ws))
            return out
        return hip.lstm_recurrence_packed(gates, packed_hh, self.cell_ws, self.h_out)

    def _seq_status_readback(self, ws):
        """Behind every one-launch recurrence: its status word travels to pinned host memory, stream-ordered and without a host
        synchronisation; `check_seq` looks at it once the copy has landed (ADVICE r3 / VERDICT r3 next 6: the word used to be read by
        nobody, and a grid that lost its compute units to another process returned NaN logits without an error)."""
        # One pinned word PER LAUNCH (a ring): a caller may enqueue many forwards before anything is looked at, and the next launch
        # clears the device-side word -- a single host word would let a later, healthy forward overwrite an earlier one's timeout.
        if self._seq_host is None:
            self._seq_host = torch.zeros(self._SEQ_RING, dtype=torch.int32).pin_memory()
            self._seq_pending = collections.deque()
        with self._seq_lock:
            if len(self._seq_pending) >= self._SEQ_RING:      # the slot about to be re-used still holds an unread word: read it first
                ev, slot = self._seq_pending.popleft()
                ev.synchronize()
                self._seq_note(slot)
            slot = self._seq_slot
            self._seq_outcome.pop(slot, None)                 # (a new launch takes the slot: its old verdict is history)
            self._seq_slot = (slot + 1) % self._SEQ_RING
            self._seq_host[slot: slot + 1].copy_(ws.view(torch.int32)[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._seq_pending.append((ev, slot))
            self._seq_last = (ev, slot)

    _SEQ_RING = 256

    def _seq_note(self, slot):
        """The verdict of the launch behind ring slot ``slot`` (its status copy has landed): latched for the next forward
        (``_seq_failed``) AND remembered per slot, so that the handle of exactly that forward still raises from its ``result()`` when
        ``check_seq`` got to the word first (it used to return NaN logits silently then)."""
        failed = int(self._seq_host[slot]) != 0
        self._seq_outcome[slot] = failed
        self._seq_failed |= failed

    def _demote(self):
        """After a resident recurrence reported a timeout: one launch per frame from now on, in the SAME arithmetic as the form that failed."""
        self.lstm_seq_mode = '0' if self.lstm_seq_mode == '1' else 'frames'
        self._tapes.clear()

    def check_seq(self, wait=False):
        """Raise if a one-launch recurrence enqueued through this plan timed out (that forward's logits hold NaN rows); the
        one-launch form is switched off for this plan then, so the caller's retry runs the per-frame launches.  ``wait``: block until
        that launch has finished (tests, `ASRModel.check`); otherwise only a finished launch is looked at -- called at the start of
        every forward, so a failure surfaces at the next call at the latest."""
        pending = self._seq_pending
        with self._seq_lock:
            while pending:
                ev, slot = pending[0]
                if wait:
                    ev.synchronize()
                elif not ev.query():
                    break
                pending.popleft()
                self._seq_note(slot)
            failed, self._seq_failed = self._seq_failed, False
        if failed:
            self._demote()
            raise hip.HipError('the one-launch LSTM recurrence of an EARLIER forward timed out waiting for its peer workgroups (the grid was '
                               'not co-resident: compute units taken by another process or stream); that forward\'s logits are invalid. '
                               'The plan now uses one launch per frame (same arithmetic): run the forward again')

    def check_seq_slot(self, ev, slot):
        """Wait for ONE one-launch recurrence (the status copy behind it) and raise if it timed out: what ``PendingLogits.result()``
        does for the forward it belongs to."""
        ev.synchronize()
        with self._seq_lock:
            try:
                self._seq_pending.remove((ev, slot))          # reported here, not again at the next forward
                failed = int(self._seq_host[slot]) != 0
            except ValueError:
                # check_seq (the next forward's poll) or the ring wrap has consumed the word already: its verdict was recorded per slot
                failed = bool(self._seq_outcome.pop(slot, False))
                if failed:
                    self._seq_failed = False                  # reported here, by the forward it belongs to
        if failed:
            self._demote()
            raise hip.HipError('the one-launch LSTM recurrence of THIS forward timed out waiting for its peer workgroups (the grid was not '
                               'co-resident: compute units taken by another process or stream); its logits are invalid. The plan now '
                               'uses one launch per frame (as NBASR_LSTM_SEQ=0): run the forward again')

    def wait_tails(self):
        """Make the current stream wait for every pipelined LSTM tail enqueued through this plan (a plain
        forward after forward_async shares the gate / cell / h buffers with the tail that is still running)."""
        cur = None
        for k, ev in enumerate(self.tail_done):
            if ev is not None:
                cur = cur or torch.cuda.current_stream(self.device)
                cur.wait_event(ev)

    def close(self):
        """Called before the plan is dropped: later allocations on the current stream may re-use its memory."""
        self.wait_tails()
        self._graphs.clear()
        self._tapes.clear()
        self._bufs.clear()
        self.side_stream = None                     # (streams.py remembers it for the main stream: a plan built again gets the same one)

    def _set_shape(self, batch, frames, use_rnn):
        from .model import FILTERS, DOWN_STRIDES, LSTM_HIDDEN
        self.batch, self.frames = batch, frames
        t, self.block_frames = frames, []
        for s in DOWN_STRIDES:
            t = (t + s - 1) // s
            self.block_frames.append(t)
        self.out_frames = self.block_frames[-1]
        pitch8 = lambda t: (t + 7) & ~7                  # noqa: E731  (bf16 rows are pitched to 8 frames: size for either storage type)
        stat_elems = max(batch * 2 * pitch8(t) for t in self.block_frames)
        self.stats = [self._buf(f'stats{i}', stat_elems) for i in range(2)]
        ws_bytes = hip.load_library().nbasr_grouped_stats_workspace_bytes(max(batch, 1), max(max(pitch8(t) for t in self.block_frames), 8), 100)
        self.stats_ws = self._buf('stats_ws', ws_bytes // 4)
        elems = max(batch * c * hip.round_up4(t) for c, t in zip(FILTERS, self.block_frames))
        self.pool = [self._buf(f'pool{i}', elems) for i in range(4)]
        self.absmax = self._buf('absmax', max(batch, 1))          # max|LayerNorm output| per utterance
        self.absmax_in = self._buf('absmax_in', max(batch, 1))    # max|model input| per utterance
        if use_rnn:
            self.gates_ws = self._buf('gates0', batch * self.out_frames * 4 * LSTM_HIDDEN)
            self.cell_ws = self._buf('cell', batch * LSTM_HIDDEN)
            self.h_out = self._buf('h_out', batch * self.out_frames * LSTM_HIDDEN)[: batch * self.out_frames * LSTM_HIDDEN] \
                .view(batch, self.out_frames, LSTM_HIDDEN)
        else:
            self.h_out = None

    def _timed(self, kind, meta, launch):
        """Run ``launch()``; when ``self.timer`` is a list, bracket it with HIP events on the current stream
        (bench.py's per-kernel roofline leg) and append (kind, meta, start, stop)."""
        if self.timer is None:
            return launch()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        out = launch()
        stop.record()
        self.timer.append((kind, meta, start, stop))
        return out

    def _cached(self, param, tag, build):
        """Derived copy of a parameter (packed / split / re-laid-out), rebuilt whenever the parameter changes.  An entry is
        valid only for the very tensor OBJECT it was built from (weak reference) at the same version: a DataParallel
        replica's freshly broadcast weight may land at the address a dead one had, with the same version counter."""
        key = (id(param), tag)
        hit = self._packed.get(key)
        if hit is None or hit[0]() is not param or hit[1] != param._version:
            if len(self._packed) >= 512:
                self._packed = {k: v for k, v in self._packed.items() if v[0]() is not None}
            if hit is not None:
                self.wait_tails()        # a pipelined tail on the side stream may still read the copy that is being replaced
            built = build()
            if self.side_stream is not None and isinstance(built, torch.Tensor):
                built.record_stream(self.side_stream)
            hit = (weakref.ref(param), param._version, built)
            self._packed[key] = hit
            self._tapes.clear()
            self._mutations += 1
        return hit[2]

    def _packed_weights(self, layer, scheme, row_tile=128):
        """Split/re-laid-out copy of a downsample conv's weights."""
        w = layer.conv.weight
        return self._cached(w, (scheme, row_tile, layer.strides), lambda: hip.pack_dense_weights(w.detach(), layer.strides, scheme, row_tile))

    def _row_tile(self, c_out, frames_out, allow_64=True):
        """Rows per workgroup of the image-path GEMM: 128, or 160 where that means less work in whole rounds of workgroups
        (a tile's cost is proportional to its rows; 256 CUs run one workgroup each).  At the benchmark shape: 160 for
        C_out = 800 (5 full row tiles instead of 7 with the last a quarter full) and 1200 (512 workgroups instead of 640)."""
        n_nt = (hip.round_up4(frames_out) + 255) // 256

        def cost(rows):
            wgs = -(-c_out // rows) * n_nt * self.batch
            return -(-wgs // 256) * rows
        rows = 160 if cost(160) < cost(128) else 128
        # smaller tiles where a small batch leaves CUs without a workgroup (8 utterances: 128 or 80 workgroups for convs 2 and 3).  A
        # smaller tile does less per operand byte staged, so it has to win by a margin: 96 rows (round 4: 13 x 16 = 208 workgroups in ONE
        # round for conv 3 at 16 utterances) by a tenth, 64 rows by a quarter (measured +4 % at 8 utterances, nothing to gain at 32)
        score = float(cost(rows))
        if allow_64:
            for r, margin in ((96, 1.1), (64, 1.25)):
                if margin * cost(r) < score:
                    rows, score = r, margin * cost(r)
        return rows

    def _bf16_tile(self, c_out, frames_out, pipe=False):
        """(rows, frames) per workgroup of the one-term bf16 GEMM: rows 128 / 160, frames 256 / 512 (results are bit-identical, the choice
        is speed only).  That flavour is bound by LDS reads; a 512-frame tile re-uses a weight fragment 8 times instead of 4 and runs
        its K-steps ~10 % faster per unit of work (measured at 32 x 1600: conv 2 312 -> 283 us, conv 3 237 -> 213) -- where the frames
        and the workgroup count still fill whole tiles and rounds (conv 1, 1600 frames = 3.1 tiles of 512: 357 -> 362).  Cost = rounds
        of 256 workgroups x tile area x that factor.  In a PIPELINED forward the 256-frame tiles stay: the wide tiles' workgroups (237-245
        registers, twice as long-lived) leave the per-frame launches of the previous batch's LSTM tail waiting for a compute unit --
        dense convs 0.983 -> 0.924 ms per forward but 9 620 -> 9 190 utterances/s (three alternating same-box runs each)."""
        forced = os.environ.get('NBASR_BF16_FTILE')
        best = None
        for rows in (128, 160):
            for ftile in ((int(forced),) if forced else (256,) if pipe else (256, 512)):
                wgs = -(-c_out // rows) * -(-hip.row_pitch(frames_out, torch.bfloat16) // ftile) * self.batch
                cost = -(-wgs // 256) * rows * ftile * (0.9 if ftile == 512 else 1.0)
                if best is None or cost < best[0]:
                    best = (cost, rows, ftile)
        return best[1], best[2]

    def _dense_tile(self, layer, frames_out):
        """(row tile, frame tile) of the image-path fp16 GEMM for this launch.  Every tiling computes the same sums in the same order
        (results are bit-identical), so the choice is speed only: `_row_tile`'s whole-rounds model, overruled by the measured table
        (dense_tile_table.json: every tiling at 4 ... 64 utterances x 1000 frames) where that knows the shape and a tiling beat the
        model's choice by more than 3 % -- 128-frame tiles and 64-row tiles at small batches (one round of workgroups: conv 3 at 8
        utterances 185 -> 168 us), 64-row tiles for conv 0 at every batch (two workgroups per CU: its ten K-steps are mostly prologue
        and epilogue)."""
        conv = layer.conv
        rows = self._row_tile(conv.out_channels, frames_out)
        best = (rows, 256)
        for (c_in, c_out, stride, t_ref), by_batch in _DENSE_TILES.items():
            if (c_in, c_out, stride) != (conv.in_channels, conv.out_channels, layer.strides) or not 0.75 * t_ref <= frames_out <= 1.34 * t_ref:
                continue
            b_ref = min(by_batch, key=lambda b: abs(math.log2(max(self.batch, 1) / b)))
            if not 0.7 * b_ref <= self.batch <= 1.42 * b_ref:
                break
            row = by_batch[b_ref]
            cand = min(row, key=row.get)
            if row[cand] < 0.97 * row.get(best, float('inf')):
                best = cand
            break
        return best

    def _packed_linear(self, linear):
        """Packed (fp16 split) copy of an nn.Linear-like weight (c_out, c_in), rebuilt whenever the parameter changes."""
        w = linear.weight if hasattr(linear, 'weight') else linear
        return self._cached(w, 'pointwise', lambda: hip.pack_pointwise_weights(w.detach()))

    def _packed_grouped(self, op):
        """[group][ci][tap][co] copy of a grouped conv's weights (what the fused cell's scalar loads read), rebuilt whenever the parameter changes."""
        w = op.conv.weight
        return self._cached(w, 'gc_wperm', lambda: hip.pack_grouped_weights(self._f32(w).contiguous(), op.groups))

    def _tail_on_side_stream(self, batch):
        """Where a pipelined forward (``forward_async``) runs its LSTM + head.  'side': on the plan's second stream, one launch per frame,
        beside the next batch's encoder (rounds 2-5).  'main': behind the encoder on the same stream, the recurrence as the resident
        XCD-local launch -- no overlap, but a recurrence of ~0.45 ms instead of 250 launches that take ~0.6 ms from the encoder they hide
        behind.  NBASR_TAIL=auto picks by batch size (measured table in DESIGN 6)."""
        if self.tail_mode == 'auto':
            return batch < self._TAIL_MAIN_FROM or self.lstm_seq_mode in ('0', '1')
        return self.tail_mode == 'side'

    _TAIL_MAIN_FROM = 1 << 30            # utterances per forward from which 'auto' keeps the tail on the main stream (set from measurements)

    def _packed_whh16(self, w):
        """w_hh as the resident operand image of the XCD-local recurrence (two fp16 terms per weight), rebuilt when the parameter changes."""
        return self._cached(w, 'whh16', lambda: hip.lstm_pack_whh16(self._f32(w).contiguous()))

    def _packed_whh(self, w):
        """Fragment-ordered copy of the LSTM's recurrent weight, rebuilt whenever the parameter changes."""
        return self._cached(w, 'whh', lambda: hip.lstm_pack_whh(self._f32(w).contiguous()))

    def _pointwise_ws(self, c_in, ld):
        need = hip.load_library().nbasr_pointwise_workspace_bytes(self.batch, c_in, ld)
        return self._buf('pointwise_ws', max(need, 16), torch.uint8)

    def _dense_part(self, c_out, ld_out):
        """Workspace for the statistics partials a dense convolution emits from its epilogue (one row pair per 16 channels)."""
        n = hip.dense_stats_part_floats(self.batch, c_out, ld_out)
        return self._buf('dense_part', n)[:n]

    def _dense(self, layer, act, act_frames, out, ln, absmax=None, blk=None, image=None, want_stats=False):
        """``absmax``: (B,) device bounds of max|act[b]| when `act` was just written by the LayerNorm kernel, else None.
        ``image`` = (image, bound): the LayerNorm of `act` was written as the pre-split operand image instead.
        ``want_stats``: the block LayerNorm behind this convolution is deferred -- the image-path kernel then emits its statistics
        partials itself (``self._conv_part``; merged in ``_norm``) instead of a statistics pass over the output."""
        if image is not None:
            self.dense_schemes[blk] = 'f16x2-image'
            b, c, ld = act.shape
            rows, ftile = self._dense_tile(layer, (act_frames + layer.strides - 1) // layer.strides)
            self.dense_row_tiles[blk], self.dense_frame_tiles[blk] = rows, ftile
            part = None
            if want_stats:
                part = self._conv_part = self._dense_part(layer.conv.out_channels, out.shape[2])
            return hip.dense_conv1d_fused_packed_f16_img(image[0], image[1], b, c, act_frames, ld,
                                                         self._packed_weights(layer, 'f16x2', rows), layer.conv.out_channels,
                                                         layer.kernel_size, layer.conv.bias.detach(), out, layer.strides, rows, part, ftile)
        if self.dense_mode != 'f32' and layer.kernel_size == 8:
            scheme = 'f16x2' if self.dense_mode == 'auto' and absmax is not None and ln is None else 'bf16x3'
            self.dense_schemes[blk] = scheme
            part = None
            if want_stats:                             # (the same 16-channel partials as the image path: the statistics do not depend on the route)
                part = self._conv_part = self._dense_part(layer.conv.out_channels, out.shape[2])
            return hip.dense_conv1d_fused_packed(act, act_frames, self._packed_weights(layer, scheme), layer.conv.out_channels,
                                                 layer.kernel_size, layer.conv.bias.detach(), (), out, layer.strides, ln, scheme,
                                                 absmax if scheme == 'f16x2' else None, part)
        self.dense_schemes[blk] = 'f32'
        return hip.dense_conv1d_fused(act, act_frames, layer.conv.weight.detach(), layer.conv.bias.detach(), (), out,
                                      layer.strides, ln, ln is not None, False)

    @staticmethod
    def _cheap_consumer(nxt):
        """Deferral pays only where normalising on load is nearly free: a following cell whose first node is a grouped
        convolution (or `zero`).  GEMM consumers (the next block's dense conv, a `linear` first node, the LSTM, the head)
        stage their input through a register pipeline where the extra per-element work costs more than the LayerNorm
        pass it saves (measured: +2 ms on the dense convs, +1.2 ms on the LSTM projection)."""
        from .model import SearchCell
        from .ops import Linear
        return isinstance(nxt, SearchCell) and not isinstance(nxt.nodes[0].op, Linear)

    def _norm(self, norm, act, act_frames, kind_meta, taps, tap_idx, nxt, out=None, defer=None):
        """LayerNorm of ``act``: returns the pending descriptor (deferred) or None after normalising in place (or into
        ``out``).  ``defer``: the caller's decision whether the consumer normalises on load (default: _cheap_consumer)."""
        if defer is None:
            defer = self._cheap_consumer(nxt)
        if not defer or out is not None:
            dst = act if out is None else out
            from .ops import PadConvRelu
            want_range = self.dense_mode == 'auto' and isinstance(nxt, PadConvRelu) and nxt.groups == 1 and nxt.kernel_size == 8
            if want_range and taps is None and out is None:
                # the consumer is the fp16-split convolution: write its pre-split operand image instead of the fp32 tensor
                # (same traffic; the convolution then gathers its tiles by LDS-DMA and does no vector staging)
                b, c, ld = act.shape
                image = self._buf('image', max(hip.load_library().nbasr_split_image_bytes(b, c, ld), 16), torch.uint8)
                self._stat_turn ^= 1
                stats = self.stats[self._stat_turn][: b * 2 * ld].view(b, 2, ld)
                bound = self.absmax[:b]
                self._timed('layernorm', kind_meta, lambda: hip.layernorm_split_image(act, norm.weight.detach(), norm.bias.detach(),
                                                                                   stats, bound, image, act_frames, norm.eps))
                self._act_image = (image, bound)
                self._act_absmax = None
                return None
            absmax = self.absmax[: act.shape[0]] if want_range else None
            self._timed('layernorm', kind_meta, lambda: hip.layernorm_channels(act, norm.weight.detach(), norm.bias.detach(),
                                                                               dst, act_frames, norm.eps, absmax))
            self._act_absmax = absmax
            return None
        self._stat_turn ^= 1
        b, _, ld = act.shape
        stats = self.stats[self._stat_turn][: b * 2 * ld].view(b, 2, ld)
        conv_part, self._conv_part = self._conv_part, None
        if conv_part is not None:
            # the convolution that wrote `act` left per-row-tile partials: merge them (a few rows per utterance) -- no pass over `act`
            c = act.shape[1]
            self._timed('stats_finalize', kind_meta, lambda: hip.grouped_stats_finalize(conv_part, stats, c, act_frames, c, norm.eps, hip.DENSE_STATS_UNIT))
        else:
            self._timed('channel_stats', kind_meta, lambda: hip.channel_stats(act, stats, act_frames, norm.eps))
        if taps is not None:                         # parity debugging: materialise a copy, the flow stays deferred
            copy = torch.empty_like(act)
            hip.layernorm_channels(act, norm.weight.detach(), norm.bias.detach(), copy, act_frames, norm.eps)
            taps[tap_idx] = copy[:, :, :act_frames].clone()
        return (stats, norm.weight.detach(), norm.bias.detach())

    def _gc_variant(self, view, node=None, ln0=None, stats=None, n_inputs=0):
        """Kernel variant of the fp32 grouped-conv node op for this launch (speed only: every variant computes the same sums in the
        same order, results are bit-identical).  NBASR_GC_F32_VARIANT=<int> forces one (diagnostics; needs ld % 8 == 0 for the
        8-frame variants; the output-split ones are never forced onto a statistics launch, which they do not have).

        Default: looked up in gc_variant_table.json, which tools/make_gc_variant_table.py derives from same-process A/B timings of
        the variants {default, output split, pipelined buffer loads, both, LDS ring, persistent LDS ring} per (taps, dilation,
        channels per group, flavour, size class) on an MI355X (tools/ubench/ab_gc_variants.py, profiles/r03_gc_variants/)."""
        forced = os.environ.get('NBASR_GC_F32_VARIANT')
        if forced is not None:
            v = int(forced)
            return (v & ~hip.GC_OSPLIT if v & hip.GC_PIPE else 0) if (v & hip.GC_OSPLIT and (stats is not None or node is None)) else v
        if node is None or not self.gc_table:
            return 0
        op = node.op
        groups = getattr(op, 'groups', 0)
        if not groups:
            return 0
        b, c, ld = view.shape
        waves_per_simd = b * groups * (-(-(ld // 4) // 64)) / 1024.0
        on_x = ln0 is not None and n_inputs == 1
        has_skips = any(type(br).__name__ == 'Identity' for br in node.branch_ops)
        flavour = ('lnx+skip' if has_skips else 'lnx') if on_x else ('skip' if has_skips else 'plain')
        head = f"{op.kernel_size},{op.dilation},{c // groups},"
        tail = f",{'small' if waves_per_simd < 4.0 else 'large'}{',stats' if stats is not None else ''}"
        v = self.gc_table.get(head + flavour + tail)
        if v is None and on_x:                           # (tables older than round 3 have one LayerNorm-on-load class)
            v = self.gc_table.get(head + 'lnx' + tail)
        return v or 0

    def _view(self, idx, channels, frames):
        ld = hip.round_up4(frames)
        return self.pool[idx][: self.batch * channels * ld].view(self.batch, channels, ld)

    # ---- whole-forward HIP graph ------------------------------------------------------------------------------------
    def _signature(self, model):
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def run_graph(self, model, x):
        """Replay the forward as ONE captured HIP graph (all ~350 launches incl. the 250 LSTM steps): removes the per-launch
        host cost and shortens the gaps between the short dependent kernels.  One graph per (batch, frames); re-captured
        when a parameter changes or a workspace grows.  Returns a tensor that the next run_graph call on the same shape
        overwrites."""
        sig = self._signature(model)
        key = (x.shape[0], x.shape[2])
        self.wait_tails()                                     # outside the capture: events of earlier pipelined calls
        hit = self._graphs.get(key)
        if hit is None or hit[1] != sig:
            self._graphs.pop(key, None)
            x_static = torch.empty(x.shape[0], x.shape[1], x.shape[2], device=self.device, dtype=x.dtype)
            x_static.copy_(x)
            for _ in range(2):                                # static initialisers, packed weights, workspace growth
                self.run(model, x_static)
            torch.cuda.synchronize(self.device)
            self.check_seq(wait=True)                         # (the warm-up forwards may have run the one-launch recurrence; the capture never does)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y_static = self.run(model, x_static, _capturing=True)
            hit = self._graphs[key] = (graph, sig, x_static, y_static)
        hit[2].copy_(x)
        hit[0].replay()
        return hit[3]

    def _ensure_pipeline(self):
        if self.side_stream is not None:
            return
        # The tail's stream must sit on another hardware queue than the stream the encoder runs on -- and than every other stream this
        # package runs chains or tails on (two chains in flight: ASRModel.forward_many) -- or the overlap silently disappears: streams.py
        # probes pool streams for that.  Normal priority (round 6; rounds 1-5: high priority, "dispatched ahead of the encoder's bulk
        # work"): measured no gain from the priority (10 528 vs 10 548 utterances/s at 64 x 1000), and a high-priority queue starves the
        # normal-priority queue it shares a pipe with -- another chain's encoder, with two chains in flight.
        # (Measured and rejected: giving the tail 8-32 compute units of its own through CU-masked streams.  The 128
        # workgroups of a step then run in several rounds and the whole pipeline slows 2-3x; tools/ubench/cu_mask_map.hip
        # documents the mask layout.)
        from . import streams
        # (a chain stream of forward_many finds its tail stream chosen already; a lone caller's first pipelined forward probes here, waiting
        # for its own stream only when it runs beside other threads of the process)
        self.side_stream = streams.tail_stream_for(self.device, torch.cuda.current_stream(self.device), whole_device=threading.active_count() == 1)
        for t in self._bufs.values():                 # allocated on the main stream, from now on also read on the side stream
            t.record_stream(self.side_stream)
        for _, _, built in self._packed.values():     # derived weights (packed w_ih / w_hh, fp32 copies): read by the tails too
            if isinstance(built, torch.Tensor):
                built.record_stream(self.side_stream)

    def _pipeline_buffers(self, channels, frames, need_enc=True, group=None):
        """Double-buffered gate pre-activations of the pipelined tail (+ an fp32 hand-over copy of the encoder output where the
        storage types differ, bf16 path): (slot, encoder output view or None)."""
        from .model import LSTM_HIDDEN
        self._ensure_pipeline()
        ld = hip.round_up4(frames)
        enc = [self._buf(f'enc_out{i}', self.batch * channels * ld) for i in range(2)] if need_enc else None
        self.gates_pipe = [self.gates_ws, self._buf('gates1', self.batch * self.out_frames * 4 * LSTM_HIDDEN)]
        if group is not None:
            # a tail group: the gates of its n forwards in ONE (frames, n * batch, 4 hidden) tensor per slot; the slot turns with the
            # group's first forward, the others write into the slot that is current
            n_floats = group[1] * self.batch * self.out_frames * 4 * LSTM_HIDDEN
            self.gates_group = [self._buf('gates_group0', n_floats), self._buf('gates_group1', n_floats)]
            if group[0] > 0:
                return self._turn, None

        def rotate():
            self._turn ^= 1
            ev = self.tail_done[self._turn]
            if ev is not None:                       # the LSTM that read these buffers two forwards ago
                torch.cuda.current_stream(self.device).wait_event(ev)
        self._host(rotate)
        k = self._turn
        return k, (enc[k][: self.batch * channels * ld].view(self.batch, channels, ld) if need_enc else None)

    # ---- launch tapes ------------------------------------------------------------------------------------------------------
    def _host(self, step, produces=False):
        """Run a host-side step of the launch sequence now and, while a tape is being recorded, note it for every replay.
        ``produces``: the step returns a freshly allocated tensor whose address later launches take (patched on replay)."""
        out = step()
        if self._recording is not None:
            self._recording.append([None, step, out.data_ptr() if produces else None, []])
        return out

    def _to_side_stream(self):
        """The launches that follow wait for everything enqueued on the current stream so far (host step of the tape)."""
        def hand_over():
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self.side_stream.wait_event(ready)
        self._host(hand_over)

    def _tail_enqueued(self, k):
        def mark():
            done = torch.cuda.Event()
            done.record(self.side_stream)
            self.tail_done[k] = done
        self._host(mark)

    def _new_logits(self, shape, dtype, pipe, numel=None):
        """Allocate the tensor the caller receives (the one per-call allocation), on the stream that writes it."""
        def alloc():
            if pipe:
                with torch.cuda.stream(self.side_stream):
                    t = torch.empty(numel if numel is not None else shape, device=self.device, dtype=dtype)
            else:
                t = torch.empty(numel if numel is not None else shape, device=self.device, dtype=dtype)
            self._tape_logits = t if numel is None else t[: shape[0] * shape[1] * shape[2]].view(shape)
            return t
        return self._host(alloc, produces=True)

    _TAPE_ENV = ('NBASR_GC_F32_VARIANT',)       # (NBASR_LSTM_SEQ and the other plan switches are read once, when the plan is built)

    def _tape_key(self, model, x, pipelined, group=None):
        """Everything the recorded launch sequence depends on; None: this call cannot use a tape."""
        if getattr(model, '_is_replica', False) or x.numel() == 0:    # DataParallel replica: its weights are fresh tensors every step
            return None
        slots, epoch = self._param_slots, _structure_epoch[0]
        if slots is None or slots[0]() is not model or slots[2] != epoch:
            found = [(m._parameters, n) for m in model.modules() for n, p in m._parameters.items() if p is not None]
            slots = self._param_slots = (weakref.ref(model), found, epoch)
        params = tuple([(d[n].data_ptr(), d[n]._version) for d, n in slots[1]])
        pipe = bool(pipelined) and model.use_rnn and self._tail_on_side_stream(x.shape[0])
        slot = -1 if not pipe else (self._turn ^ 1) if (group is None or group[0] == 0) else self._turn      # (a group's first forward rotates the slot)
        return (x.dtype, tuple(x.shape), x.data_ptr() % 16 == 0, bool(pipelined), slot, group,
                torch.cuda.current_stream(self.device).cuda_stream, tuple(os.environ.get(k) for k in self._TAPE_ENV),
                epoch, tuple(map(id, model.model)), params, self._structure_fingerprint(model))

    @staticmethod
    def _structure_fingerprint(model):
        """The PLAIN attributes the launch sequence reads on every call and that no registration hook sees:
        ``node.branch_ops`` is a python list, ``cell.use_norm`` / ``op.dilation`` / ``norm.eps`` / ``model.use_rnn`` are plain
        attributes -- mutating one after a tape exists must not replay the old sequence."""
        fp = [model.use_rnn, model.training, getattr(model, 'dropout_rate', 0.0)]
        for layer in model.model:
            nodes = getattr(layer, 'nodes', None)
            if nodes is not None:
                fp.append(layer.use_norm)
                fp.append(layer.norm_layer.eps if layer.use_norm else None)
                for node in nodes:
                    op = node.op
                    fp.append((type(op), getattr(op, 'kernel_size', 0), getattr(op, 'dilation', 0), getattr(op, 'groups', 0),
                               tuple(map(type, node.branch_ops))))
            else:
                fp.append((type(layer), getattr(layer, 'eps', None), getattr(layer, 'kernel_size', 0), getattr(layer, 'strides', 0),
                           getattr(layer, 'groups', 0), getattr(layer, 'p', None)))
        return tuple(fp)

    def run(self, model, x, taps=None, pipelined=False, _capturing=False, group=None):
        """``_enqueue`` + the hand-over of a one-launch recurrence's status slot to the handle a pipelined forward returns.
        ``group`` = (g, n): this pipelined forward is the g-th of a TAIL GROUP of n forwards of the same shape enqueued back to back
        through this plan -- their LSTM + head run as ONE recurrence over n x batch utterances behind the last of them."""
        self._seq_last = None
        out = self._enqueue(model, x, taps, pipelined, _capturing, group)
        if self._seq_last is not None and isinstance(out, PendingLogits):
            out._seq = (self,) + self._seq_last
        return out

    def _enqueue(self, model, x, taps=None, pipelined=False, _capturing=False, group=None):
        """Enqueue one forward of ``model`` (its parameters are read now, so a DataParallel replica runs with its own).
        ``taps`` (a dict) receives a copy of every layer's output, keyed by the layer's index in ``model.model``, in the
        oracle's layouts ((B,C,T) for encoder layers and the LSTM).  ``pipelined``: LSTM + head go to the side stream and
        a ``PendingLogits`` is returned."""
        if x.device != self.device:
            raise hip.HipError(f'input on {x.device}, plan on {self.device}')
        if not _capturing and self._seq_pending and not torch.cuda.is_current_stream_capturing():
            self.check_seq()                           # (an event query is not a capturable operation)
        wdtype = model.model[0].conv.weight.dtype          # (not model.parameters(): a DataParallel replica has none)
        if x.dtype != wdtype:
            raise hip.HipError(f'input is {x.dtype} but the model\'s parameters are {wdtype}: cast one of them '
                               f'(model.to(torch.bfloat16) / x.bfloat16() for the bf16 path)')
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise hip.HipError(f'input must be float32 or bfloat16 (got {x.dtype})')
        x = x.detach().contiguous()
        group = self._group_or_none(model, x, taps, pipelined, _capturing, group)
        if group is not None:
            body = lambda m, xx, tp, pl, cp: self._run_f32(m, xx, tp, pl, cp, group)          # noqa: E731
        else:
            body = self._run_bf16 if x.dtype == torch.bfloat16 else self._run_f32
        key = None
        if self.tape_mode and taps is None and not _capturing and self.timer is None:
            key = self._tape_key(model, x, pipelined, group)
        if key is None:
            return body(model, x, taps, pipelined, _capturing)
        tape = self._tapes.get(key)
        if tape is not None:
            self.tape_replays += 1
            return tape.replay(self, x)
        if len(self._tape_seen) > 256:
            self._tape_seen.clear()
        seen = self._tape_seen[key] = self._tape_seen.get(key, 0) + 1
        if seen < 2:
            # first sight of this key: workspaces may grow and derived weights may be built -- not a sequence worth keeping
            return body(model, x, taps, pipelined, _capturing)
        entries, before = [], self._mutations
        self._recording = entries
        hip.start_tape(entries)
        try:
            out = body(model, x, taps, pipelined, _capturing)
        finally:
            hip.stop_tape()
            self._recording = None
        if self._mutations == before:
            if len(self._tapes) >= 48:               # (a tail group of 8 on two slots: 16 tapes of one shape)
                self._tapes.clear()
            self._tapes[key] = LaunchTape(entries, x.data_ptr(), bool(pipelined),
                                          bool(pipelined) and model.use_rnn and self._tail_on_side_stream(x.shape[0]), group)
        self._tape_logits = None
        return out

    def _group_or_none(self, model, x, taps, pipelined, capturing, group):
        """The tail group this forward can really run in: fp32, a pipelined tail on the side stream, the fp16-pair per-frame recurrence
        and the packed input projection (the defaults), n x batch utterances within the recurrence's workspace form."""
        if group is None or group[1] <= 1 or not pipelined or taps is not None or capturing or x.dtype != torch.float32:
            return None
        from .model import LSTM_HIDDEN
        if not (model.use_rnn and self._tail_on_side_stream(x.shape[0]) and self.linear_mode == 'f16x2' and self.lstm_seq_mode in ('auto', 'frames')
                and isinstance(model.model[-1], torch.nn.Linear) and isinstance(model.model[-2], torch.nn.LSTM)
                and hip.lstm_xcd_workspace_bytes(group[1] * x.shape[0], LSTM_HIDDEN)):
            return None
        return (int(group[0]), int(group[1]))

    def _group_member(self, group, logits_all):
        """The handle of the forward just enqueued as member ``group`` = (g, n); the last member hands every handle its rows."""
        g, n = group
        if g == 0:
            self._group_handles = []
        handle = GroupLogits()
        self._group_handles.append(handle)
        if g == n - 1:
            if len(self._group_handles) != n or logits_all is None:
                raise hip.HipError(f'tail group of {n}: {len(self._group_handles)} forwards were enqueued through this plan before its last one')
            rows, done = logits_all.shape[0] // n, self.tail_done[self._turn]
            for j, h in enumerate(self._group_handles):
                h._set(logits_all[j * rows:(j + 1) * rows], done)
            self._group_handles = []
        return handle

    def _run_f32(self, model, x, taps, pipelined, _capturing, group=None):
        from .model import SearchCell
        from .ops import PadConvRelu
        import torch.nn as nn
        pipe = bool(pipelined) and model.use_rnn and taps is None and self._tail_on_side_stream(x.shape[0])
        if not pipe and not _capturing:
            # the plain path shares the gate / cell / h buffers with pipelined tails that may still be running
            self.wait_tails()
        self._set_shape(x.shape[0], x.shape[2], model.use_rnn)
        act, act_frames, cur = x, self.frames, None      # `cur`: pool index holding `act` (None: caller's x)
        if self.dense_mode != 'f32' and (x.shape[-1] % 4 or x.data_ptr() % 16):
            # the packed dense conv fetches aligned 4-frame quads: bring a ragged-length input into the pitched layout
            cur = 2
            act = hip.repitch(x, self._view(cur, x.shape[1], self.frames), self.frames)
        pending = None                                   # (stats, gamma, beta) when `act` still awaits its LayerNorm
        self._act_absmax = None                          # set by _norm when it wrote `act` together with max|act[b]|
        self._act_image = None                           # set by _norm when it wrote the LayerNorm of `act` as the conv's image
        input_range = None
        if self.dense_mode == 'auto' and act.shape[0] > 0:
            # the model input is caller data: one small reduction gives the first conv its range AND decides, per utterance
            # and on the device, whether that range is tame enough for the scaled fp16 scheme (nbasr.h: nbasr_input_range)
            input_range = hip.input_range(act, act_frames, self._buf('input_range', 4 * act.shape[0])[: 4 * act.shape[0]])
        pipe_k, tail_ctx = None, None
        self._stat_turn = 0
        lin_ctx = (self._packed_linear, self._pointwise_ws) if self.linear_mode == 'f16x2' else None
        blk = -1
        logits = None
        n_layers = len(model.model)
        for idx, layer in enumerate(model.model):
            if isinstance(layer, PadConvRelu):
                blk += 1
                dst = 0 if cur != 0 else 1
                t_out = self.block_frames[blk]
                out = self._view(dst, layer.conv.out_channels, t_out)
                ln, src, src_frames, amax, blk_now, img = pending, act, act_frames, self._act_absmax, blk, self._act_image
                meta = (blk, layer.conv.in_channels, layer.conv.out_channels, layer.kernel_size, t_out, 0)
                # the block LayerNorm behind this convolution is deferred into the next cell's load: its statistics then come out of
                # the convolution's own epilogue (round 5) instead of a pass over the output
                nxt1 = model.model[idx + 1] if idx + 1 < n_layers else None
                nxt2 = model.model[idx + 2] if idx + 2 < n_layers else None
                # (the finalize kernel merges <= 128 partial rows: wider than 128 x 16 channels falls back to the statistics pass)
                want_stats = (self.conv_stats and isinstance(nxt1, nn.LayerNorm) and self._cheap_consumer(nxt2) and layer.kernel_size == 8
                              and -(-layer.conv.out_channels // hip.DENSE_STATS_UNIT) <= 128)
                self._conv_part = None
                if input_range is not None and layer.kernel_size == 8 and ln is None and img is None:
                    rng, input_range = input_range, None
                    # fp16 split with per-utterance fall-back to bf16x3 (extreme / non-finite input); image path: one split
                    # pass over the input, then the same DMA-only GEMM as convs 1-3
                    bi, ci, ldi = src.shape
                    image = self._buf('input_image', max(hip.load_library().nbasr_split_image_bytes(bi, ci, ldi), 16), torch.uint8)
                    rows, ftile = self._dense_tile(layer, t_out)
                    self.dense_row_tiles[blk], self.dense_frame_tiles[blk] = rows, ftile
                    self.dense_schemes[blk] = 'f16x2' if image is None else 'f16x2-image'
                    w16 = self._packed_weights(layer, 'f16x2', rows) if image is not None else self._packed_weights(layer, 'f16x2')
                    part = None
                    if want_stats:                         # (both legs -- fp16 image, bf16x3 for extreme utterances -- write their utterances' partials)
                        part = self._conv_part = self._dense_part(layer.conv.out_channels, out.shape[2])
                    self._timed('dense_conv', meta, lambda: hip.dense_conv1d_first_ranged(
                        src, src_frames, rng, w16, self._packed_weights(layer, 'bf16x3'),
                        layer.conv.out_channels, layer.kernel_size, layer.conv.bias.detach(), out, layer.strides, image, rows, part, ftile))
                else:
                    input_range = None
                    self._timed('dense_conv', meta, lambda: self._dense(layer, src, src_frames, out, ln, amax, blk_now, img, want_stats))
                act, act_frames, cur, pending, self._act_absmax, self._act_image = out, t_out, dst, None, None, None
                if taps is not None:
                    taps[idx] = self._tap(act, act_frames)
            elif isinstance(layer, nn.LayerNorm):
                if act.dim() != 3 or pending is not None:
                    raise RuntimeError('LayerNorm in an unexpected position of the layer list')
                pending = self._norm(layer, act, act_frames, (blk, act.shape[1], act.shape[1], 0, act_frames, 0), taps, idx,
                                     model.model[idx + 1] if idx + 1 < n_layers else None)
                if taps is not None and pending is None:
                    taps[idx] = self._tap(act, act_frames)
            elif isinstance(layer, SearchCell):
                free = [i for i in range(4) if i != cur]
                if len(layer.nodes) > len(free):
                    raise NotImplementedError(f'cells with {len(layer.nodes)} nodes need a larger buffer pool')
                nxt = model.model[idx + 1] if idx + 1 < n_layers else None
                feeds_tail = pipe and isinstance(nxt, (nn.Dropout, nn.LSTM))
                # the LSTM's input projection pre-splits its operand in a streaming pass that applies a pending LayerNorm while
                # loading (gemm_pointwise_split.hip), and it runs on the MAIN stream in both modes: the cell in front of it defers
                # its LayerNorm like a cell in front of a grouped conv does -- no materialised copy of the encoder output
                after = model.model[idx + 2] if isinstance(nxt, nn.Dropout) and idx + 2 < n_layers else nxt
                defer = self._cheap_consumer(nxt) or (isinstance(after, nn.LSTM) and self.linear_mode == 'f16x2')
                # a deferred cell LayerNorm whose producer is a grouped conv gets its statistics from that node's
                # epilogue (no statistics pass over the tensor)
                last_op = layer.nodes[-1].op
                cell_gpp = (hip.grouped_cell_fits(layer.filters, hip.round_up4(act_frames), last_op.groups)
                            if (self.cell_fusion and len(layer.nodes) == 3
                                and all(isinstance(n.op, PadConvRelu) and n.op.groups > 1 for n in layer.nodes)) else 0)
                fused = cell_gpp > 0
                epilogue_stats = (layer.use_norm and defer
                                  and isinstance(last_op, PadConvRelu) and last_op.groups > 1)
                if fused:
                    mask = 0
                    for bit, (j, i) in enumerate(((0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2))):
                        if type(layer.nodes[j].branch_ops[i]).__name__ == 'Identity':
                            mask |= 1 << bit
                    view = self._view(free[2], layer.filters, act_frames)
                    specs = [(self._packed_grouped(n.op), n.op.conv.bias.detach(), n.op.kernel_size, n.op.dilation) for n in layer.nodes]
                    n_skips = [sum(type(br).__name__ == 'Identity' for br in n.branch_ops) for n in layer.nodes]
                    meta = (blk, layer.filters, tuple(sp[2] for sp in specs), tuple(n_skips), act_frames, 0)
                    src, ln0 = act, pending
                    if epilogue_stats:
                        self._stat_turn ^= 1
                        ld = view.shape[2]
                        new_stats = self.stats[self._stat_turn][: self.batch * 2 * ld].view(self.batch, 2, ld)
                    cell_ws = self.stats_ws if epilogue_stats else None
                    self._timed('grouped_cell', meta, lambda: hip.grouped_cell_fused(src, specs, mask, view, act_frames, last_op.groups, ln0, cell_ws))
                    outs = [act, None, None, view]
                if not fused:
                    outs = [act]
                for j, (node, dst) in enumerate(zip(layer.nodes, free) if not fused else ()):
                    n_skips = sum(type(br).__name__ == 'Identity' for br in node.branch_ops)
                    kind = {'PadConvRelu': 'grouped_conv', 'Linear': 'linear_op', 'Zero': 'skip_sum'}[type(node.op).__name__]
                    meta = (blk, layer.filters, layer.filters, getattr(node.op, 'kernel_size', 1), act_frames, n_skips)
                    view = self._view(dst, layer.filters, act_frames)
                    ln0, st = pending, None
                    if epilogue_stats and j == len(layer.nodes) - 1:
                        self._stat_turn ^= 1
                        ld = view.shape[2]
                        new_stats = self.stats[self._stat_turn][: self.batch * 2 * ld].view(self.batch, 2, ld)
                        st = (new_stats, self.stats_ws, layer.norm_layer.eps)
                    outs.append(self._timed(kind, meta, lambda: node_into(node, outs, act_frames, view, ln0, st, lin_ctx, self._gc_variant(view, node, ln0, st, len(outs)))))
                act, cur, pending = outs[-1], free[len(layer.nodes) - 1], None
                if feeds_tail:
                    pipe_k, _ = self._pipeline_buffers(layer.filters, act_frames, need_enc=False, group=group)
                if epilogue_stats:
                    norm = layer.norm_layer
                    self._timed('stats_finalize', (blk, layer.filters, layer.filters, 0, act_frames, 0),
                                lambda: hip.grouped_stats_finalize(self.stats_ws, new_stats, layer.filters, act_frames,
                                                                   last_op.groups, norm.eps, cell_gpp if fused else 4))
                    pending = (new_stats, norm.weight.detach(), norm.bias.detach())
                    if taps is not None:
                        copy = torch.empty_like(act)
                        hip.layernorm_channels(act, norm.weight.detach(), norm.bias.detach(), copy, act_frames, norm.eps)
                        taps[idx] = copy[:, :, :act_frames].clone()
                elif layer.use_norm:
                    pending = self._norm(layer.norm_layer, act, act_frames, (blk, layer.filters, layer.filters, 0, act_frames, 0),
                                         taps, idx, nxt, None, defer)
                if taps is not None and pending is None:
                    taps[idx] = self._tap(act, act_frames)
            elif isinstance(layer, nn.Dropout):
                if taps is not None:                      # identity: eval mode or p == 0 (checked by the model)
                    taps[idx] = taps[idx - 1]
            elif isinstance(layer, nn.LSTM):
                ln, src, src_frames = pending, act, act_frames
                w_ih, w_hh = layer.weight_ih_l0.detach(), layer.weight_hh_l0.detach()
                b_ih, b_hh = layer.bias_ih_l0.detach(), layer.bias_hh_l0.detach()
                gates = self.gates_pipe[pipe_k] if pipe else self.gates_ws
                if group is not None:
                    # TAIL GROUP (round 6): this forward's gates become utterances g * batch .. of the group's (frames, n * batch, 4 hidden)
                    # gate tensor; the last member runs ONE recurrence and ONE head over all of them -- a frame of the recurrence costs
                    # at 32 utterances what it costs at 8, and no utterance's h depends on the batch it is computed in (bit-identical)
                    g, n = group
                    gb = n * self.batch
                    gates_all = self.gates_group[pipe_k]
                    packed_ih, ws = self._packed_linear(layer.weight_ih_l0), self._pointwise_ws(src.shape[1], src.shape[2])
                    self._timed('lstm_projection', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_input_projection_packed(src, src_frames, packed_ih, b_ih, b_hh, gates_all, layer.hidden_size,
                                                                         ws, ln, batch_total=gb, batch_offset=g * self.batch))
                    if g < n - 1:
                        return self._group_member(group, None)
                    self._to_side_stream()
                    tail_ctx = torch.cuda.stream(self.side_stream)
                    tail_ctx.__enter__()
                    hidden = layer.hidden_size
                    xcd_ws = self._buf('lstm_xcd_group', hip.lstm_xcd_workspace_bytes(gb, hidden), torch.uint8)
                    cell_all = self._buf('cell_group', gb * hidden)[: gb * hidden]
                    h_all = self._buf('h_out_group', gb * act_frames * hidden)[: gb * act_frames * hidden].view(gb, act_frames, hidden)
                    gview = gates_all[: act_frames * gb * 4 * hidden].view(act_frames, gb, 4 * hidden)
                    self._timed('lstm', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_recurrence_frames16(gview, self._packed_whh16(layer.weight_hh_l0), cell_all, h_all, xcd_ws))
                    head = model.model[idx + 1]
                    logits = self._new_logits((gb, act_frames, head.out_features), torch.float32, True)
                    hip.linear_head(h_all, head.weight.detach(), head.bias.detach(), logits)
                    self._tail_enqueued(pipe_k)
                    tail_ctx.__exit__(None, None, None)
                    return self._group_member(group, logits)
                # the input projection is one large GEMM: it stays with the encoder; only the recurrence moves over
                if self.linear_mode == 'f16x2':
                    packed_ih, ws = self._packed_linear(layer.weight_ih_l0), self._pointwise_ws(src.shape[1], src.shape[2])
                    self._timed('lstm_projection', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_input_projection_packed(src, src_frames, packed_ih, b_ih, b_hh, gates,
                                                                         layer.hidden_size, ws, ln))
                else:
                    self._timed('lstm_projection', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_input_projection(src, src_frames, w_ih, b_ih, b_hh, gates, layer.hidden_size, ln))
                if pipe:                                   # everything from here on runs on the side stream
                    self._to_side_stream()
                    tail_ctx = torch.cuda.stream(self.side_stream)
                    tail_ctx.__enter__()
                self._timed('lstm', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                            lambda: self._recurrence(gates, layer.weight_hh_l0, layer.hidden_size, pipe, _capturing))
                act, pending = self.h_out, None            # (batch, frames, hidden)
                if taps is not None:
                    taps[idx] = self._tap(act, act_frames)
            elif isinstance(layer, nn.Linear):
                logits = self._new_logits((self.batch, act_frames, layer.out_features), torch.float32, tail_ctx is not None)
                if act is self.__dict__.get('h_out'):
                    hip.linear_head(act, layer.weight.detach(), layer.bias.detach(), logits)
                else:
                    hip.linear_head_bct(act, act_frames, layer.weight.detach(), layer.bias.detach(), logits, pending)
                act, pending = logits, None
            else:
                raise TypeError(f'unsupported layer {type(layer).__name__} in the model list')
        if tail_ctx is not None:
            self._tail_enqueued(pipe_k)
            tail_ctx.__exit__(None, None, None)
        if idx != n_layers - 1 or logits is None:
            raise RuntimeError('the model list does not end in the CTC head')
        if pipelined:
            return PendingLogits(logits, self.tail_done[pipe_k] if tail_ctx is not None else None)
        if taps is not None:
            taps[len(model.model) - 1] = logits.clone()
        return logits

    # ---- bf16 storage path (BASELINE config 4) ----------------------------------------------------------------------------
    def _f32(self, param):
        """fp32 copy of a (bf16) parameter, rebuilt when it changes: the kernels read weights of the node ops, biases,
        gamma / beta and the LSTM / head matrices as fp32 whatever the storage type of the activations (exact: every bf16
        value is an fp32 value)."""
        if param.dtype == torch.float32:
            return param.detach()
        return self._cached(param, 'f32', lambda: param.detach().float().contiguous())

    def _view16(self, idx, channels, frames):
        ld = hip.row_pitch(frames, torch.bfloat16)
        return self.pool16[idx][: self.batch * channels * ld].view(self.batch, channels, ld)

    def _run_bf16(self, model, x, taps, pipelined, capturing):
        """The forward with activations and GEMM operands STORED as bfloat16 (what `model.to(torch.bfloat16)(x.bfloat16())`
        is in the reference): dense convs as one bf16 MFMA per product on a producer-written operand image, node ops /
        LayerNorm reading and writing bf16 rows (half the HBM bytes) with fp32 arithmetic and ONE rounding per tensor,
        LayerNorm statistics, LSTM gates / state and the head in fp32; logits returned as bfloat16."""
        from .model import SearchCell, FILTERS, LSTM_HIDDEN, LN_EPS
        from .ops import PadConvRelu, Linear, Zero, Identity
        import torch.nn as nn
        bf16 = torch.bfloat16
        pipe = bool(pipelined) and model.use_rnn and taps is None and self._tail_on_side_stream(x.shape[0])
        if not pipe and not capturing:
            self.wait_tails()
        self._set_shape(x.shape[0], x.shape[2], model.use_rnn)
        B = self.batch
        lds = [hip.row_pitch(t, bf16) for t in self.block_frames]
        elems = max(B * c * ld for c, ld in zip(FILTERS, lds))
        self.pool16 = [self._buf(f'pool16_{i}', elems, bf16) for i in range(4)]
        lib = hip.load_library()

        def variant_for(frames):
            # 8 frames per lane (16-byte accesses) where rows are long; the narrow blocks run better as more, lighter waves
            # (tools/bench_gc_variants.py on an MI355X: 40 vs 49 us at 1 600 frames, 39 vs 37 us at 400)
            return (hip.GC_FPL8 | hip.GC_WPERM) if frames >= 512 else hip.GC_WPERM

        def image_of(act, frames, norm=None, stats=None, eps=0.0):
            b, c, ld = act.shape
            img = self._buf('image16', max(lib.nbasr_bf16_image_bytes(b, c, ld), 16), torch.uint8)
            return hip.bf16_image(act, img, frames, norm, stats, eps)

        def grouped_weight(op, variant):
            w = op.conv.weight
            if variant & hip.GC_WPERM:
                return self._cached(w, 'gc_wperm', lambda: hip.pack_grouped_weights(self._f32(w).contiguous(), op.groups))
            return self._f32(w)

        # model input -> pitched bf16 rows (whole 16-byte chunks)
        act, act_frames, cur = x, self.frames, None
        if x.shape[-1] % 8 or x.data_ptr() % 16:
            cur = 2
            act = hip.repitch(x, self._view16(cur, x.shape[1], self.frames), self.frames)
        pending, image = None, None            # deferred LayerNorm of `act` / operand image holding (the LayerNorm of) `act`
        self._stat_turn = 0
        pipe_k, tail_ctx, logits = None, None, None
        blk, n_layers = -1, len(model.model)
        act_is_f32 = False                     # set once the encoder output has been handed over to the fp32 tail
        for idx, layer in enumerate(model.model):
            nxt = model.model[idx + 1] if idx + 1 < n_layers else None
            if isinstance(layer, PadConvRelu):
                blk += 1
                if pending is not None:
                    raise RuntimeError('a dense convolution cannot take a deferred LayerNorm in the bf16 path')
                if image is None:
                    image = image_of(act, act_frames)
                dst = 0 if cur != 0 else 1
                t_out = self.block_frames[blk]
                out = self._view16(dst, layer.conv.out_channels, t_out)
                rows, ftile = self._bf16_tile(layer.conv.out_channels, t_out, pipe)
                self.dense_row_tiles[blk], self.dense_frame_tiles[blk] = rows, ftile
                w = layer.conv.weight
                packed = self._cached(w, ('bf16', rows, layer.strides),
                                      lambda: hip.pack_dense_weights_bf16(self._f32(w), layer.strides, rows))
                bias, src_shape, img_now, frames_now = self._f32(layer.conv.bias), act.shape, image, act_frames
                self.dense_schemes[blk] = 'bf16'
                self._timed('dense_conv', (blk, layer.conv.in_channels, layer.conv.out_channels, layer.kernel_size, t_out, 0),
                            lambda: hip.dense_conv1d_bf16_img(img_now, B, src_shape[1], frames_now, src_shape[2], packed,
                                                              layer.conv.out_channels, layer.kernel_size, bias, out, layer.strides, rows, ftile))
                act, act_frames, cur, image = out, t_out, dst, None
                if taps is not None:
                    taps[idx] = act[:, :, :act_frames].clone()
            elif isinstance(layer, (nn.LayerNorm, SearchCell)):
                if isinstance(layer, SearchCell):
                    free = [i for i in range(4) if i != cur]
                    if len(layer.nodes) > len(free):
                        raise NotImplementedError(f'cells with {len(layer.nodes)} nodes need a larger buffer pool')
                    last_op = layer.nodes[-1].op
                    norm = layer.norm_layer if layer.use_norm else None
                    after_cell = model.model[idx + 2] if isinstance(nxt, nn.Dropout) and idx + 2 < n_layers else nxt
                    epilogue_stats = (norm is not None and (self._cheap_consumer(nxt) or isinstance(after_cell, nn.LSTM))
                                      and isinstance(last_op, PadConvRelu) and last_op.groups > 1)
                    outs = [act]
                    # three grouped convs: ONE launch where a row fits a workgroup (grouped_cell.hip; x1 and x2 rounded to bf16 exactly
                    # where the node launches store them, so the result is the same bit for bit)
                    cell_gpp = (hip.grouped_cell_fits(layer.filters, act.shape[2], last_op.groups)
                                if (self.cell_fusion and len(layer.nodes) == 3
                                    and all(isinstance(n.op, PadConvRelu) and n.op.groups > 1 for n in layer.nodes)) else 0)
                    if cell_gpp:
                        mask = 0
                        for bit, (j, i) in enumerate(((0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2))):
                            if isinstance(layer.nodes[j].branch_ops[i], Identity):
                                mask |= 1 << bit
                        view = self._view16(free[2], layer.filters, act_frames)
                        specs = [(None, self._f32(n.op.conv.bias), n.op.kernel_size, n.op.dilation) for n in layer.nodes]
                        n_sk = [sum(isinstance(br, Identity) for br in n.branch_ops) for n in layer.nodes]
                        meta = (blk, layer.filters, tuple(sp[2] for sp in specs), tuple(n_sk), act_frames, 0)
                        src, ln0, cell_ws = act, pending, (self.stats_ws if epilogue_stats else None)
                        if self.cell_mfma and hip.grouped_cell_mfma_fits(layer.filters, act.shape[2], last_op.groups):
                            # every tensor of the bf16 model is a bfloat16 tensor: the products go to the matrix cores unchanged (grouped_cell_mfma.hip;
                            # 133 against 281 us for a 32 x 800 x 1600 cell).  That kernel has no statistics by-product: a consumer that
                            # normalises on load gets them from a read pass over the result (~25 us)
                            groups, epilogue_stats = last_op.groups, False
                            mspecs = [(self._cached(n.op.conv.weight, 'cell_mfma', (lambda w=n.op.conv.weight: hip.grouped_cell_mfma_pack(self._f32(w), groups))),
                                       sp[1], sp[2], sp[3]) for n, sp in zip(layer.nodes, specs)]
                            self._timed('grouped_cell_mfma', meta, lambda: hip.grouped_cell_mfma(src, mspecs, mask, view, act_frames, groups, ln0))
                        else:
                            vspecs = [(self._packed_grouped(n.op), sp[1], sp[2], sp[3]) for n, sp in zip(layer.nodes, specs)]
                            self._timed('grouped_cell', meta, lambda: hip.grouped_cell_fused(src, vspecs, mask, view, act_frames, last_op.groups, ln0, cell_ws))
                        outs = [act, None, None, view]
                    for j, (node, dst) in enumerate(zip(layer.nodes, free) if not cell_gpp else ()):
                        if len(outs) != len(node.branch_ops):
                            raise AssertionError('Branch op and input list have different lenghts')
                        skips = [src for br, src in zip(node.branch_ops, outs) if isinstance(br, Identity)]
                        n_skips = len(skips)
                        on_x = pending is not None and len(outs) == 1
                        on_s0 = pending is not None and isinstance(node.branch_ops[0], Identity)
                        ln = pending if (on_x or on_s0) else None
                        view, op, last = self._view16(dst, layer.filters, act_frames), node.op, outs[-1]
                        meta = (blk, layer.filters, layer.filters, getattr(op, 'kernel_size', 1), act_frames, n_skips)
                        if isinstance(op, PadConvRelu):
                            ws = self.stats_ws if (epilogue_stats and j == len(layer.nodes) - 1) else None
                            variant = variant_for(act_frames)
                            wt, bs = grouped_weight(op, variant), self._f32(op.conv.bias)
                            self._timed('grouped_conv', meta, lambda: hip.grouped_conv1d_node(
                                last, wt, bs, skips, view, act_frames, op.groups, op.kernel_size, op.dilation, ln, on_x, on_s0, ws, variant))
                        elif isinstance(op, Zero):
                            self._timed('skip_sum', meta, lambda: hip.skip_sum(skips, view, act_frames, ln if on_s0 else None, on_s0))
                        elif isinstance(op, Linear):
                            # round 4: one bf16 MFMA per product on a bf16 operand image (gemm_pointwise_bf16.hip); rounds 2-3 bridged the op
                            # through the fp32 split GEMM (x -> fp32, two fp16 terms, three MFMAs, y -> bf16, a separate skip sum)
                            b_, c_, ld_ = last.shape
                            wl = op.linear.weight
                            packed = self._cached(wl, 'pointwise_bf16', lambda: hip.pack_pointwise_weights_bf16(self._f32(wl)))
                            ws16 = self._buf('pointwise_bf16_ws', max(lib.nbasr_pointwise_bf16_workspace_bytes(b_, c_, ld_), 16), torch.uint8)
                            self._timed('linear_op', meta, lambda: hip.linear_fused_bf16(
                                last, act_frames, packed, op.linear.out_features, self._f32(op.linear.bias), skips, view, ws16, ln, on_x, on_s0))
                        else:
                            raise TypeError(f'unsupported node operation {type(op).__name__}')
                        outs.append(view)
                    act, cur, pending = outs[-1], free[len(layer.nodes) - 1], None
                else:
                    norm, epilogue_stats, cell_gpp = layer, False, 0
                    if act.dim() != 3 or pending is not None:
                        raise RuntimeError('LayerNorm in an unexpected position of the layer list')
                feeds_tail = isinstance(nxt, (nn.Dropout, nn.LSTM, nn.Linear))
                after = model.model[idx + 2] if isinstance(nxt, nn.Dropout) and idx + 2 < n_layers else nxt
                meta = (blk, act.shape[1], act.shape[1], 0, act_frames, 0)
                if feeds_tail and isinstance(after, nn.LSTM):
                    # the LSTM's input projection takes the bf16 encoder output itself (round 4: bf16 operand image, one MFMA per
                    # product) and applies a pending LayerNorm while writing that image: statistics only, no fp32 copy of the tensor
                    if pipe:
                        pipe_k, _ = self._pipeline_buffers(act.shape[1], act.shape[2], need_enc=False)
                    if norm is not None:
                        g32, b32 = self._f32(norm.weight), self._f32(norm.bias)
                        self._stat_turn ^= 1
                        b, c, ld = act.shape
                        stats = self.stats[self._stat_turn][: b * 2 * ld].view(b, 2, ld)
                        src = act
                        if epilogue_stats:
                            self._timed('stats_finalize', meta, lambda: hip.grouped_stats_finalize(self.stats_ws, stats, c, act_frames,
                                                                                                 last_op.groups, norm.eps, cell_gpp or 4))
                        else:
                            self._timed('channel_stats', meta, lambda: hip.channel_stats(src, stats, act_frames, norm.eps))
                        pending = (stats, g32, b32)
                        if taps is not None:                 # parity debugging: materialise a copy, the flow stays deferred
                            copy = torch.empty_like(act)
                            hip.layernorm_channels(act, g32, b32, copy, act_frames, norm.eps)
                            taps[idx] = copy[:, :, :act_frames].clone()
                    elif taps is not None:
                        taps[idx] = act[:, :, :act_frames].clone()
                elif feeds_tail:
                    # hand-over to the fp32 tail (LSTM projection / head): the last LayerNorm writes fp32
                    b, c, ld = act.shape
                    if pipe:
                        pipe_k, enc = self._pipeline_buffers(c, ld)           # fp32 (B, C, ld) double buffer
                    else:
                        enc = self._buf('enc32', b * c * ld)[: b * c * ld].view(b, c, ld)
                    src = act
                    if norm is not None:
                        g32, b32 = self._f32(norm.weight), self._f32(norm.bias)
                        self._timed('layernorm', meta, lambda: hip.layernorm_channels(src, g32, b32, enc, act_frames, norm.eps))
                    else:
                        hip.convert(src, enc)
                    act, cur, act_is_f32 = enc, None, True
                    if taps is not None:
                        taps[idx] = act[:, :, :act_frames].clone()
                elif norm is None:
                    if taps is not None:
                        taps[idx] = act[:, :, :act_frames].clone()
                else:
                    g32, b32 = self._f32(norm.weight), self._f32(norm.bias)
                    self._stat_turn ^= 1
                    b, c, ld = act.shape
                    stats = self.stats[self._stat_turn][: b * 2 * ld].view(b, 2, ld)
                    src = act
                    if isinstance(nxt, PadConvRelu):
                        # the consumer is a dense conv: the LayerNorm writes its operand image (statistics on the way)
                        self._timed('layernorm', meta, lambda: image_of(src, act_frames, (g32, b32), stats, norm.eps))
                        image = self._bufs['image16']
                        if taps is not None:
                            copy = torch.empty_like(act)
                            hip.layernorm_channels(act, g32, b32, copy, act_frames, norm.eps)
                            taps[idx] = copy[:, :, :act_frames].clone()
                    elif self._cheap_consumer(nxt):
                        if epilogue_stats:
                            self._timed('stats_finalize', meta, lambda: hip.grouped_stats_finalize(self.stats_ws, stats, c, act_frames,
                                                                                                 last_op.groups, norm.eps, cell_gpp or 4))
                        else:
                            self._timed('channel_stats', meta, lambda: hip.channel_stats(src, stats, act_frames, norm.eps))
                        pending = (stats, g32, b32)
                        if taps is not None:
                            copy = torch.empty_like(act)
                            hip.layernorm_channels(act, g32, b32, copy, act_frames, norm.eps)
                            taps[idx] = copy[:, :, :act_frames].clone()
                    else:
                        self._timed('layernorm', meta, lambda: hip.layernorm_channels(src, g32, b32, src, act_frames, norm.eps))
                        if taps is not None:
                            taps[idx] = act[:, :, :act_frames].clone()
            elif isinstance(layer, nn.Dropout):
                if taps is not None:
                    taps[idx] = taps[idx - 1]
            elif isinstance(layer, nn.LSTM):
                src, src_frames = act, act_frames
                b_ih, b_hh = self._f32(layer.bias_ih_l0), self._f32(layer.bias_hh_l0)
                gates = self.gates_pipe[pipe_k] if pipe else self.gates_ws
                w_ih32, w_hh32 = self._f32(layer.weight_ih_l0), self._f32(layer.weight_hh_l0)
                if act_is_f32:
                    # (an fp32 hand-over copy of the encoder output, should a caller have made one: the fp16-split GEMM of rounds 2-3)
                    packed_ih = self._cached(layer.weight_ih_l0, 'pointwise', lambda: hip.pack_pointwise_weights(w_ih32))
                    ws = self._pointwise_ws(src.shape[1], src.shape[2])
                    self._timed('lstm_projection', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_input_projection_packed(src, src_frames, packed_ih, b_ih, b_hh, gates, layer.hidden_size, ws, None))
                else:
                    packed_ih = self._cached(layer.weight_ih_l0, 'pointwise_bf16', lambda: hip.pack_pointwise_weights_bf16(w_ih32))
                    ws16 = self._buf('pointwise_bf16_ws', max(lib.nbasr_pointwise_bf16_workspace_bytes(B, src.shape[1], src.shape[2]), 16), torch.uint8)
                    ln_x, pending = pending, None
                    self._timed('lstm_projection', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_input_projection_bf16(src, src_frames, packed_ih, b_ih, b_hh, gates, layer.hidden_size, ws16, ln_x))
                if pipe:
                    self._to_side_stream()
                    tail_ctx = torch.cuda.stream(self.side_stream)
                    tail_ctx.__enter__()
                self._timed('lstm', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                            lambda: self._recurrence(gates, layer.weight_hh_l0, layer.hidden_size, pipe, capturing))
                act = self.h_out
                if taps is not None:
                    taps[idx] = act.permute(0, 2, 1).clone()
            elif isinstance(layer, nn.Linear):
                n = B * act_frames * layer.out_features
                n8 = (n + 7) & ~7
                logits32 = self._buf('logits32', max(n8, 8))[: max(n8, 8)]      # tails run one after another on the side stream
                l32 = logits32[:n].view(B, act_frames, layer.out_features)
                w32, b32 = self._f32(layer.weight), self._f32(layer.bias)
                if act is self.h_out:
                    hip.linear_head(act, w32, b32, l32)
                else:
                    hip.linear_head_bct(act, act_frames, w32, b32, l32, None)
                # (the conversion works in 8-element chunks: the last chunk's padding is converted too and never returned)
                out16 = self._new_logits((B, act_frames, layer.out_features), bf16, tail_ctx is not None, numel=max(n8, 8))
                hip.convert(logits32, out16)
                logits = out16[:n].view(B, act_frames, layer.out_features)
                act = logits
            else:
                raise TypeError(f'unsupported layer {type(layer).__name__} in the model list')
        if tail_ctx is not None:
            self._tail_enqueued(pipe_k)
            tail_ctx.__exit__(None, None, None)
        if logits is None:
            raise RuntimeError('the model list does not end in the CTC head')
        if pipelined:
            return PendingLogits(logits, self.tail_done[pipe_k] if tail_ctx is not None else None)
        if taps is not None:
            taps[len(model.model) - 1] = logits.clone()
        return logits

    def _tap(self, act, frames):
        if act is self.__dict__.get('h_out'):
            return act.permute(0, 2, 1).clone()              # (B, H, T) like the oracle's LSTM tap
        return act[:, :, :frames].clone()


class PlanPool:
    """Idle ``ForwardPlan``s of one model, per device.  ``acquire`` hands a plan to exactly one caller at a time, so
    concurrent forwards (``torch.nn.DataParallel`` worker threads -- one per device, or several on one device) never share
    workspaces, while back-to-back calls from one thread keep re-using the same plan (LIFO).  Replicas made by
    ``nn.Module._replicate_for_data_parallel`` share this object (shallow ``__dict__`` copy); that is fine because a plan
    holds no reference to a model -- the model is an argument of every run."""

    def __init__(self):
        self._lock = threading.Lock()
        self._idle = {}              # device index -> [ForwardPlan]

    def __reduce__(self):
        return (PlanPool, ())            # copies / pickles of a model start with an empty pool

    def acquire(self, device):
        """An idle plan for a forward on ``device``'s CURRENT stream.  A plan is handed back (``release``) as soon as its forward has been
        ENQUEUED: its workspaces are still being read and written by that stream.  The same stream takes it again for free (stream
        order); another stream (two threads running forwards on two streams of one device: round 6, `bench.py --in-flight`) gets a plan
        of its own if it can -- the launch tapes of a plan are per stream as well -- and otherwise the least recently released one,
        behind the event ``release`` recorded and behind that plan's pipelined tails (round 6: before, such a plan was re-used at once
        and the two streams raced on its workspaces)."""
        device = torch.device(device)
        stream = torch.cuda.current_stream(device) if device.type == 'cuda' and torch.cuda.is_available() else None
        sid = stream.cuda_stream if stream is not None else None
        with self._lock:
            idle = self._idle.get(device.index)
            if idle:
                for i in range(len(idle) - 1, -1, -1):                  # most recently released first
                    if idle[i]._released_on == sid or idle[i]._released_on is None:
                        return idle.pop(i)
                if len(idle) >= self.MAX_IDLE_PER_DEVICE:               # bounded: a caller that makes a fresh stream per forward must not grow the pool for ever
                    plan = idle.pop(0)
                    if plan._released_event is not None:
                        stream.wait_event(plan._released_event)
                    plan.wait_tails()
                    return plan
        return ForwardPlan(device)

    MAX_IDLE_PER_DEVICE = 6

    def release(self, plan):
        if plan.device.type == 'cuda' and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            stream = torch.cuda.current_stream(plan.device)
            if plan._released_event is None:
                plan._released_event = torch.cuda.Event()
            plan._released_event.record(stream)
            plan._released_on = stream.cuda_stream
        with self._lock:
            self._idle.setdefault(plan.device.index, []).append(plan)

    def poll(self, device):
        """Look (without waiting) at the status words of the one-launch recurrences enqueued through EVERY idle plan of ``device`` and
        raise for a finished one that timed out: with several plans per device (concurrent callers) the failing plan may not be the
        one the next forward is handed."""
        device = torch.device(device)
        with self._lock:
            for plan in self._idle.get(device.index, ()):
                if plan._seq_pending:
                    plan.check_seq()

    def values(self):
        """Idle plans, most recently used last (bench.py / tests read the last run's bookkeeping from them)."""
        with self._lock:
            return [p for idle in self._idle.values() for p in idle]

    def __len__(self):
        return len(self.values())

    def clear(self):
        with self._lock:
            plans = [p for idle in self._idle.values() for p in idle]
            self._idle = {}
        for p in plans:
            if torch.cuda.is_available():
                with torch.cuda.device(p.device):
                    p.close()
