"""Host-side executor: walks the model's layer list and enqueues the HIP kernels.

Replaces the isinstance-dispatch loop of the reference's ``ASRModel.forward`` (model.py:116-131),
``SearchCell.forward`` (model.py:49-59) and ``Node.forward`` (model.py:13-22):

* activations stay in (batch, channels, frames) order in pitched workspace buffers (row pitch =
  frames rounded up to 4, pitch columns kept at zero), so none of the reference's permutes, padded
  copies, ``zeros_like`` branches or ``0 + x`` adds exist here;
* a node is ONE launch: main op + bias + ReLU + clamp + the sum of its flagged skip inputs;
* LayerNorm runs on the channel (slow) dimension in place.

Four workspace buffers sized for the widest block are rotated; nothing is allocated per call
except the returned logits.
"""
import os
import weakref

import torch

from . import hip


def node_into(node, inputs, frames, out, ln0=None, stats=None, linear_ctx=None):
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
        hip.grouped_conv1d_fused(last, op.conv.weight.detach(), op.conv.bias.detach(), skips, out, frames,
                                 op.groups, op.kernel_size, op.dilation, ln, on_x, on_s0, None,
                                 stats[1] if stats is not None else None, 0.0)
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

    def result(self):
        if self._event is not None:
            cur = torch.cuda.current_stream(self._logits.device)
            cur.wait_event(self._event)
            self._logits.record_stream(cur)
            self._event = None
        return self._logits


class ForwardPlan:
    """Workspace + launch sequence of one model for one (batch, frames, device)."""

    def __init__(self, model, batch, frames, device):
        from .model import FILTERS, DOWN_STRIDES, LSTM_HIDDEN
        self._model = weakref.ref(model)
        self.batch, self.frames, self.device = batch, frames, device
        t, self.block_frames = frames, []
        for s in DOWN_STRIDES:
            t = (t + s - 1) // s
            self.block_frames.append(t)
        self.out_frames = self.block_frames[-1]
        self.timer = None
        # dense k=8 convs, all fp32-accurate:
        #   'auto' (default) = 2-way fp16 split (3 MFMAs per product) wherever the input has just been written by the
        #                      LayerNorm kernel (which also emits the per-utterance max|x| the scheme's range scaling
        #                      needs), the 3-way bf16 split (6 MFMAs per product, fp32's exponent range) elsewhere;
        #   'bf16x3' = 3-way bf16 split everywhere;  'f32' = the exact-fp32 MFMA kernel
        self.dense_mode = os.environ.get('NBASR_DENSE_MODE', 'auto')
        if self.dense_mode not in ('auto', 'bf16x3', 'f32'):
            raise ValueError(f'NBASR_DENSE_MODE must be auto, bf16x3 or f32, got {self.dense_mode!r}')
        self._packed = {}            # (id(layer), scheme) -> (weight key, packed tensor)
        # per-frame linear maps (`linear` node ops, LSTM input projection): 'f16x2' = fp16 matrix cores with pre-split
        # activations (default), 'f32' = the exact-fp32 MFMA GEMM
        self.linear_mode = os.environ.get('NBASR_LINEAR_MODE', 'f16x2')
        if self.linear_mode not in ('f16x2', 'f32'):
            raise ValueError(f'NBASR_LINEAR_MODE must be f16x2 or f32, got {self.linear_mode!r}')
        self._pw_ws = None           # scratch of the pre-split activation image, grown on demand
        # LayerNorm -> dense conv hand-off as a pre-split fp16 image (NBASR_IMAGE_MODE=0: fp32 tensor + in-GEMM staging)
        self.image_mode = os.environ.get('NBASR_IMAGE_MODE', '1') != '0'
        # row tile of the image-path GEMM: auto (per layer, see _row_tile) | 128 | 160
        self.row_tile_mode = os.environ.get('NBASR_ROW_TILE', 'auto')
        if self.row_tile_mode not in ('auto', '128', '160'):
            raise ValueError(f'NBASR_ROW_TILE must be auto, 128 or 160, got {self.row_tile_mode!r}')
        self._image = None
        self._act_image = None
        self.absmax = torch.zeros(max(batch, 1), device=device, dtype=torch.float32)   # max|LayerNorm output| per utterance
        self.absmax_in = torch.zeros(max(batch, 1), device=device, dtype=torch.float32)   # max|model input| per utterance
        self.dense_schemes = {}      # block -> scheme used by the last run (read by bench.py)
        self.dense_row_tiles = {}    # block -> rows per workgroup of the image-path GEMM in the last run
        # LayerNorm: 'deferred' = one statistics pass, consumers normalise while loading (default);
        # 'materialize' = the stand-alone LayerNorm kernel writes the normalised tensor
        self.ln_mode = os.environ.get('NBASR_LN_MODE', 'deferred')
        if self.ln_mode not in ('deferred', 'materialize'):
            raise ValueError(f'NBASR_LN_MODE must be deferred or materialize, got {self.ln_mode!r}')
        self.epilogue_stats = os.environ.get('NBASR_EPILOGUE_STATS', '1') != '0'
        # opt-in: cells whose three nodes are grouped convs run as ONE launch (x1, x2 stay in LDS).  Bit-identical, but at
        # B=64/T=1000 it is VALU/latency-bound (45-50 TFLOP/s at 2-3 waves per SIMD): 139/263/210/164 us per cell vs
        # 177/264/159/117 us for three HBM-bound launches -- a win only in block 0, so it is off by default
        self.cell_fusion = os.environ.get('NBASR_CELL_FUSION', '0') == '1'
        stat_elems = max(batch * 2 * hip.round_up4(t) for t in self.block_frames)
        self.stats = [torch.empty(max(stat_elems, 4), device=device, dtype=torch.float32) for _ in range(2)]
        self.stats_ws = hip.grouped_stats_workspace(batch, max(hip.round_up4(t) for t in self.block_frames), 100, device)
        elems = max(batch * c * hip.round_up4(t) for c, t in zip(FILTERS, self.block_frames))
        self.pool = [torch.empty(max(elems, 4), device=device, dtype=torch.float32) for _ in range(4)]
        if model.use_rnn:
            self.gates_ws = torch.empty(batch * self.out_frames * 4 * LSTM_HIDDEN, device=device, dtype=torch.float32)
            self.cell_ws = torch.empty(batch * LSTM_HIDDEN, device=device, dtype=torch.float32)
            self.h_out = torch.empty(batch, self.out_frames, LSTM_HIDDEN, device=device, dtype=torch.float32)
        # pipelined mode (forward_async): the latency-bound LSTM + head of batch i run on a side stream while the main
        # stream already runs the encoder of batch i+1; the encoder output is double-buffered for that
        self.side_stream = None
        self._graph, self._graph_sig = None, None
        self.enc_out = None
        self.tail_done = [None, None]
        self._turn = 0

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

    def _packed_weights(self, layer, scheme, row_tile=128):
        """Split/re-laid-out copy of a downsample conv's weights, rebuilt whenever the parameter changes."""
        w = layer.conv.weight
        key = (w.data_ptr(), w._version)
        hit = self._packed.get((id(layer), scheme, row_tile))
        if hit is None or hit[0] != key:
            hit = (key, hip.pack_dense_weights(w.detach(), layer.strides, scheme, row_tile))
            self._packed[(id(layer), scheme, row_tile)] = hit
        return hit[1]

    def _row_tile(self, c_out, frames_out):
        """Rows per workgroup of the image-path GEMM: 128, or 160 where that means less work in whole rounds of workgroups
        (a tile's cost is proportional to its rows; 256 CUs run one workgroup each).  At the benchmark shape: 160 for
        C_out = 800 (5 full row tiles instead of 7 with the last a quarter full) and 1200 (512 workgroups instead of 640)."""
        if self.row_tile_mode != 'auto':
            return int(self.row_tile_mode)
        n_nt = (hip.round_up4(frames_out) + 255) // 256

        def cost(rows):
            wgs = -(-c_out // rows) * n_nt * self.batch
            return -(-wgs // 256) * rows
        return 160 if cost(160) < cost(128) else 128

    def _packed_linear(self, linear):
        """Packed (fp16 split) copy of an nn.Linear-like weight (c_out, c_in), rebuilt whenever the parameter changes."""
        w = linear.weight if hasattr(linear, 'weight') else linear
        key = (w.data_ptr(), w._version)
        hit = self._packed.get((id(w), 'pointwise'))
        if hit is None or hit[0] != key:
            hit = (key, hip.pack_pointwise_weights(w.detach()))
            self._packed[(id(w), 'pointwise')] = hit
        return hit[1]

    def _packed_whh(self, w):
        """Fragment-ordered copy of the LSTM's recurrent weight, rebuilt whenever the parameter changes."""
        key = (w.data_ptr(), w._version)
        hit = self._packed.get((id(w), 'whh'))
        if hit is None or hit[0] != key:
            hit = (key, hip.lstm_pack_whh(w.detach()))
            self._packed[(id(w), 'whh')] = hit
        return hit[1]

    def _pointwise_ws(self, c_in, ld):
        need = hip.load_library().nbasr_pointwise_workspace_bytes(self.batch, c_in, ld)
        if self._pw_ws is None or self._pw_ws.numel() < need:
            self._pw_ws = hip.pointwise_workspace(self.batch, c_in, ld, self.device)
        return self._pw_ws

    def _dense(self, layer, act, act_frames, out, ln, absmax=None, blk=None, image=None):
        """``absmax``: (B,) device bounds of max|act[b]| when `act` was just written by the LayerNorm kernel, else None.
        ``image`` = (image, bound): the LayerNorm of `act` was written as the pre-split operand image instead."""
        if image is not None:
            self.dense_schemes[blk] = 'f16x2-image'
            b, c, ld = act.shape
            rows = self.dense_row_tiles[blk] = self._row_tile(layer.conv.out_channels, (act_frames + layer.strides - 1) // layer.strides)
            return hip.dense_conv1d_fused_packed_f16_img(image[0], image[1], b, c, act_frames, ld,
                                                         self._packed_weights(layer, 'f16x2', rows), layer.conv.out_channels,
                                                         layer.kernel_size, layer.conv.bias.detach(), out, layer.strides, rows)
        if self.dense_mode != 'f32' and layer.kernel_size == 8:
            scheme = 'f16x2' if self.dense_mode == 'auto' and absmax is not None and ln is None else 'bf16x3'
            self.dense_schemes[blk] = scheme
            return hip.dense_conv1d_fused_packed(act, act_frames, self._packed_weights(layer, scheme), layer.conv.out_channels,
                                                 layer.kernel_size, layer.conv.bias.detach(), (), out, layer.strides, ln, scheme,
                                                 absmax if scheme == 'f16x2' else None)
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

    def _norm(self, norm, act, act_frames, kind_meta, taps, tap_idx, nxt, out=None):
        """LayerNorm of ``act``: returns the pending descriptor (deferred) or None after normalising in place (or into
        ``out``)."""
        if self.ln_mode == 'materialize' or not self._cheap_consumer(nxt) or out is not None:
            dst = act if out is None else out
            from .ops import PadConvRelu
            want_range = self.dense_mode == 'auto' and isinstance(nxt, PadConvRelu) and nxt.groups == 1 and nxt.kernel_size == 8
            if want_range and self.image_mode and taps is None and out is None:
                # the consumer is the fp16-split convolution: write its pre-split operand image instead of the fp32 tensor
                # (same traffic; the convolution then gathers its tiles by LDS-DMA and does no vector staging)
                b, c, ld = act.shape
                need = hip.load_library().nbasr_split_image_bytes(b, c, ld)
                if self._image is None or self._image.numel() < need:
                    self._image = hip.split_image(b, c, ld, self.device)
                self._stat_turn ^= 1
                stats = self.stats[self._stat_turn][: b * 2 * ld].view(b, 2, ld)
                bound = self.absmax[:b]
                self._timed('layernorm', kind_meta, lambda: hip.layernorm_split_image(act, norm.weight.detach(), norm.bias.detach(),
                                                                                   stats, bound, self._image, act_frames, norm.eps))
                self._act_image = (self._image, bound)
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
        self._timed('channel_stats', kind_meta, lambda: hip.channel_stats(act, stats, act_frames, norm.eps))
        if taps is not None:                         # parity debugging: materialise a copy, the flow stays deferred
            copy = torch.empty_like(act)
            hip.layernorm_channels(act, norm.weight.detach(), norm.bias.detach(), copy, act_frames, norm.eps)
            taps[tap_idx] = copy[:, :, :act_frames].clone()
        return (stats, norm.weight.detach(), norm.bias.detach())

    def _view(self, idx, channels, frames):
        ld = hip.round_up4(frames)
        return self.pool[idx][: self.batch * channels * ld].view(self.batch, channels, ld)

    # ---- whole-forward HIP graph ------------------------------------------------------------------------------------
    def _signature(self, model):
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def run_graph(self, x):
        """Replay the forward as ONE captured HIP graph (all ~350 launches incl. the 250 LSTM steps): removes the per-launch
        host cost and shortens the gaps between the short dependent kernels.  The graph is re-captured when a parameter
        changes.  Returns a tensor that the next run_graph call overwrites."""
        model = self._model()
        sig = self._signature(model)
        if self._graph is None or self._graph_sig != sig:
            self._graph = None
            self._x_static = torch.empty(self.batch, x.shape[1], self.frames, device=self.device, dtype=torch.float32)
            self._x_static.copy_(x)
            for _ in range(2):                                # static initialisers, packed weights, allocator warm-up
                self.run(self._x_static)
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._y_static = self.run(self._x_static)
            self._graph, self._graph_sig = graph, sig
        self._x_static.copy_(x)
        self._graph.replay()
        return self._y_static

    def _ensure_pipeline(self):
        if self.side_stream is not None:
            return
        from .model import FILTERS
        # high priority: its own hardware queue (an ordinary second stream can end up sharing the main stream's queue,
        # e.g. once RCCL has created its streams, and the overlap silently disappears), and the short dependent LSTM
        # steps get dispatched ahead of the encoder's bulk work.
        # (Measured and rejected: giving the tail 8-32 compute units of its own through CU-masked streams.  The 128
        # workgroups of a step then run in several rounds and the whole pipeline slows 2-3x; tools/ubench/cu_mask_map.hip
        # documents the mask layout.)
        self.side_stream = torch.cuda.Stream(device=self.device, priority=-1)
        ld = hip.round_up4(self.out_frames)
        self.enc_out = [torch.empty(self.batch, FILTERS[-1], ld, device=self.device, dtype=torch.float32) for _ in range(2)]
        self.gates_pipe = [self.gates_ws, torch.empty_like(self.gates_ws)]

    def _pipeline_buffers(self, channels, frames):
        self._ensure_pipeline()
        if tuple(self.enc_out[0].shape[1:]) != (channels, hip.round_up4(frames)):
            raise RuntimeError('unexpected encoder output shape for the pipelined tail')
        self._turn ^= 1
        k = self._turn
        if self.tail_done[k] is not None:            # the LSTM that read these buffers two forwards ago
            torch.cuda.current_stream(self.device).wait_event(self.tail_done[k])
        return k, self.enc_out[k]

    def run(self, x, taps=None, pipelined=False):
        """Enqueue one forward.  ``taps`` (a dict) receives a copy of every layer's output, keyed by the
        layer's index in ``model.model``, in the oracle's layouts ((B,C,T) for encoder layers and the LSTM).
        ``pipelined``: LSTM + head go to the side stream and a ``PendingLogits`` is returned."""
        from .model import SearchCell
        from .ops import PadConvRelu
        import torch.nn as nn

        model = self._model()
        if model is None:
            raise RuntimeError('the model this plan belongs to no longer exists')
        if x.dtype != torch.float32:
            raise hip.HipError(f'input must be float32 (got {x.dtype})')
        x = x.detach().contiguous()
        act, act_frames, cur = x, self.frames, None      # `cur`: pool index holding `act` (None: caller's x)
        if self.dense_mode != 'f32' and (x.shape[-1] % 4 or x.data_ptr() % 16):
            # the packed dense conv fetches aligned 4-frame quads: bring a ragged-length input into the pitched layout
            cur = 2
            act = hip.repitch(x, self._view(cur, x.shape[1], self.frames), self.frames)
        pending = None                                   # (stats, gamma, beta) when `act` still awaits its LayerNorm
        self._act_absmax = None                          # set by _norm when it wrote `act` together with max|act[b]|
        self._act_image = None                           # set by _norm when it wrote the LayerNorm of `act` as the conv's image
        if self.dense_mode == 'auto' and act.shape[0] > 0:
            # the model input is unbounded: one small reduction gives the first conv its range, too
            self._act_absmax = hip.absmax(act, self.absmax_in[: act.shape[0]])
        pipe = bool(pipelined) and model.use_rnn and taps is None
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
                self._timed('dense_conv', (blk, layer.conv.in_channels, layer.conv.out_channels, layer.kernel_size, t_out, 0),
                            lambda: self._dense(layer, src, src_frames, out, ln, amax, blk_now, img))
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
                # a deferred cell LayerNorm whose producer is a grouped conv gets its statistics from that node's
                # epilogue (no statistics pass over the tensor)
                last_op = layer.nodes[-1].op
                fused = (self.cell_fusion and len(layer.nodes) == 3
                         and all(isinstance(n.op, PadConvRelu) and n.op.groups > 1 for n in layer.nodes)
                         and hip.grouped_cell_fits(layer.filters, hip.round_up4(act_frames), last_op.groups))
                if fused:
                    mask = 0
                    for bit, (j, i) in enumerate(((0, 0), (1, 0), (1, 1), (2, 0), (2, 1), (2, 2))):
                        if type(layer.nodes[j].branch_ops[i]).__name__ == 'Identity':
                            mask |= 1 << bit
                    view = self._view(free[2], layer.filters, act_frames)
                    specs = [(n.op.conv.weight.detach(), n.op.conv.bias.detach(), n.op.kernel_size, n.op.dilation) for n in layer.nodes]
                    n_skips = [sum(type(br).__name__ == 'Identity' for br in n.branch_ops) for n in layer.nodes]
                    meta = (blk, layer.filters, tuple(sp[2] for sp in specs), tuple(n_skips), act_frames, 0)
                    src, ln0 = act, pending
                    self._timed('grouped_cell', meta, lambda: hip.grouped_cell_fused(src, specs, mask, view, act_frames, last_op.groups, ln0))
                    outs = [act, None, None, view]
                epilogue_stats = (not fused and self.epilogue_stats and layer.use_norm and self.ln_mode == 'deferred' and self._cheap_consumer(nxt) and not feeds_tail
                                  and isinstance(last_op, PadConvRelu) and last_op.groups > 1)
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
                    outs.append(self._timed(kind, meta, lambda: node_into(node, outs, act_frames, view, ln0, st, lin_ctx)))
                act, cur, pending = outs[-1], free[len(layer.nodes) - 1], None
                if feeds_tail:
                    pipe_k, enc = self._pipeline_buffers(layer.filters, act_frames)
                if epilogue_stats:
                    norm = layer.norm_layer
                    self._timed('stats_finalize', (blk, layer.filters, layer.filters, 0, act_frames, 0),
                                lambda: hip.grouped_stats_finalize(self.stats_ws, new_stats, layer.filters, act_frames,
                                                                   last_op.groups, norm.eps))
                    pending = (new_stats, norm.weight.detach(), norm.bias.detach())
                    if taps is not None:
                        copy = torch.empty_like(act)
                        hip.layernorm_channels(act, norm.weight.detach(), norm.bias.detach(), copy, act_frames, norm.eps)
                        taps[idx] = copy[:, :, :act_frames].clone()
                elif layer.use_norm:
                    pending = self._norm(layer.norm_layer, act, act_frames, (blk, layer.filters, layer.filters, 0, act_frames, 0),
                                         taps, idx, nxt, enc if feeds_tail else None)
                    if feeds_tail:
                        act, cur = enc, None
                elif feeds_tail:
                    hip.repitch(act, enc, act_frames)
                    act, cur = enc, None
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
                    ready = torch.cuda.Event()
                    ready.record(torch.cuda.current_stream(self.device))
                    self.side_stream.wait_event(ready)
                    tail_ctx = torch.cuda.stream(self.side_stream)
                    tail_ctx.__enter__()
                if os.environ.get('NBASR_LSTM_UNPACKED') == '1':       # diagnostics: the (4H, H)-layout step kernel
                    self._timed('lstm', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_recurrence(gates, w_hh, self.cell_ws, self.h_out))
                else:
                    packed_hh = self._packed_whh(layer.weight_hh_l0)
                    self._timed('lstm', (blk, layer.input_size, layer.hidden_size, 0, act_frames, 0),
                                lambda: hip.lstm_recurrence_packed(gates, packed_hh, self.cell_ws, self.h_out))
                act, pending = self.h_out, None            # (batch, frames, hidden)
                if taps is not None:
                    taps[idx] = self._tap(act, act_frames)
            elif isinstance(layer, nn.Linear):
                logits = torch.empty(self.batch, act_frames, layer.out_features, device=self.device, dtype=torch.float32)
                if act.dim() == 3 and act is self.__dict__.get('h_out'):
                    hip.linear_head(act, layer.weight.detach(), layer.bias.detach(), logits)
                else:
                    hip.linear_head_bct(act, act_frames, layer.weight.detach(), layer.bias.detach(), logits, pending)
                act, pending = logits, None
            else:
                raise TypeError(f'unsupported layer {type(layer).__name__} in the model list')
        if tail_ctx is not None:
            done = torch.cuda.Event()
            done.record(self.side_stream)
            tail_ctx.__exit__(None, None, None)
            self.tail_done[pipe_k] = done
        if idx != n_layers - 1 or logits is None:
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
