"""The differentiable path: one ``torch.autograd.Function`` per op of the model, forward AND backward through the
C ABI, and ``model_forward`` which strings them together -- what ``ASRModel.forward`` runs in training mode with gradients enabled,
so that the reference's ``loss.backward()`` (trainer.py:220-223) reaches every parameter.

* ``grouped_pad_conv_relu``  reference ``ops.PadConvRelu`` with groups > 1 (ops.py:24-30): vector-ALU dgrad, MFMA wgrad
* ``dense_pad_conv_relu``    the dense k = 8 downsample convs (model.py:82-89) and the per-frame ``linear`` op (ops.py:42-50)
* ``layer_norm_channels``    ``nn.LayerNorm`` over the channel dimension of (B, C, T) (model.py:55-58)
* ``lstm``                   the single-layer LSTM (model.py:100,118-121), BPTT
* ``pointwise_linear``       the CTC head (model.py:101-103,122-124)
* ``skip_sum``               a node's sum over its flagged inputs (model.py:22)

Correctness first: one launch group per op, nothing fused or deferred, tensor re-layouts as torch copies; gradients are checked against
the reference modules' own autograd (tests/golden/grad_fixtures.npz, tests/test_backward_gpu.py) and, for the whole model, against fp64
autograd through the oracle (tests/test_model_gpu.py).  Training-mode dropout (p > 0) is ATen's dropout on each op's output, as the
reference places it (ops.py:22,29).  The fused inference executor (executor.py) stays the fast path for eval() / torch.no_grad().
"""
import os

import torch

from . import hip


def _pitched(t):
    frames = t.shape[2]
    ld = hip.round_up4(frames)
    t = t.detach()
    if ld == frames and t.is_contiguous():
        return t, frames
    buf = torch.empty(t.shape[0], t.shape[1], ld, device=t.device, dtype=t.dtype)
    hip.repitch(t.contiguous(), buf, frames)
    return buf, frames


class _GroupedPadConvRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, kernel, dilation):
        xp, frames = _pitched(x)
        z = torch.empty_like(xp)
        hip.grouped_conv1d_fused(xp, weight.detach(), bias.detach(), (), z, frames, groups, kernel, dilation)
        ctx.save_for_backward(xp, weight.detach(), z)
        ctx.cfg = (frames, groups, kernel, dilation)
        return z[:, :, :frames]

    @staticmethod
    def backward(ctx, dz):
        xp, weight, z = ctx.saved_tensors
        frames, groups, kernel, dilation = ctx.cfg
        dzp, _ = _pitched(dz)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = hip.grouped_conv1d_backward(xp, weight, z, dzp, frames, groups, kernel, dilation, need_dx, need_dw)
        return (dx[:, :, :frames] if dx is not None else None), dw, db, None, None, None


class _LayerNormChannels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xp, frames = _pitched(x)
        stats = torch.empty(xp.shape[0], 2, xp.shape[2], device=xp.device, dtype=torch.float32)
        hip.channel_stats(xp, stats, frames, eps)
        y = torch.empty_like(xp)
        hip.layernorm_channels(xp, gamma.detach(), beta.detach(), y, frames, eps)
        ctx.save_for_backward(xp, stats, gamma.detach())
        ctx.frames = frames
        return y[:, :, :frames]

    @staticmethod
    def backward(ctx, dy):
        xp, stats, gamma = ctx.saved_tensors
        dyp, _ = _pitched(dy)
        dx, dgamma, dbeta = hip.layernorm_channels_backward(xp, stats, gamma, dyp, ctx.frames)
        return dx[:, :, :ctx.frames], dgamma, dbeta, None


class _DensePadConvRelu(torch.autograd.Function):
    """Dense k = 8 downsample conv (stride 1 | 2) or the per-frame `linear` op (k = 1) with ReLU and clamp; forward of the k = 8 convs on the
    split 16-bit GEMMs (NBASR_DENSE_MODE = auto (default: two-term fp16 split) | bf16x3 | f32: the exact-fp32 MFMA GEMM), backward through hip.dense_conv1d_backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, normalized_input=False):
        xp, frames = _pitched(x)
        kernel = weight.shape[2] if weight.dim() == 3 else 1
        w3 = weight.detach() if weight.dim() == 3 else weight.detach().unsqueeze(-1)
        t_out = (frames + stride - 1) // stride
        y = torch.empty(xp.shape[0], weight.shape[0], hip.round_up4(t_out), device=xp.device, dtype=xp.dtype)
        mode = os.environ.get('NBASR_DENSE_MODE', 'auto')
        if kernel == 8 and mode == 'bf16x3' and xp.data_ptr() % 16 == 0:
            # the three-term bf16 split (fp32's range, no range information needed, fp32-level error): half the time of the exact-fp32
            # MFMA GEMM; the weights change every step, so they are packed per call (tens of microseconds)
            hip.dense_conv1d_fused_packed(xp, frames, hip.pack_dense_weights(w3.contiguous(), stride, 'bf16x3'), weight.shape[0], 8, bias.detach(),
                                          (), y, stride, None, 'bf16x3')
        elif kernel == 8 and mode != 'f32' and xp.data_ptr() % 16 == 0:
            # what the inference executor runs for the model's first conv, for all four: a range summary of the input per utterance, its
            # pre-split fp16 image, the LDS-DMA-only two-term fp16 GEMM (3 MFMAs per fp32 product, against 6 for bf16x3), and the bf16x3
            # leg for utterances the summary calls extreme (non-finite samples, > 2^12 of dynamic range) -- routed on the device
            wc = w3.contiguous()
            rng = torch.empty(xp.shape[0], 4, device=xp.device, dtype=torch.float32)
            hip.input_range(xp, frames, rng)
            if normalized_input:
                # a LayerNorm's output: only the non-finite flag routes (the inference executor scales these by max |x| alone as well);
                # the quiet-frame rule is for caller data, and would send utterances with a near-constant frame to the slower leg
                rng[:, 1] = rng[:, 0]
            hip.dense_conv1d_first_ranged(xp, frames, rng, hip.pack_dense_weights(wc, stride, 'f16x2', 128), hip.pack_dense_weights(wc, stride, 'bf16x3'),
                                          weight.shape[0], 8, bias.detach(), y, stride, hip.split_image(*xp.shape, xp.device), 128)
        else:
            hip.dense_conv1d_fused(xp, frames, w3, bias.detach(), (), y, stride)
        ctx.save_for_backward(xp, weight.detach(), y)
        ctx.cfg = (frames, stride, t_out, kernel)
        return y[:, :, :t_out]

    @staticmethod
    def backward(ctx, dy):
        xp, weight, y = ctx.saved_tensors
        frames, stride, t_out, _ = ctx.cfg
        dyp, _ = _pitched(dy)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = hip.dense_conv1d_backward(xp, weight, y, dyp, frames, stride, need_dx, need_dw)
        return (dx[:, :, :frames] if dx is not None else None), dw, db, None, None


class _LSTM(torch.autograd.Function):
    """nn.LSTM(C, H), one layer, unidirectional, zero initial state (reference model.py:100,118-121): x (B, C, T) -> h (B, T, H)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        xp, frames = _pitched(x)
        b, hidden = xp.shape[0], w_hh.shape[1]
        gates = torch.empty(max(frames, 1), b, 4 * hidden, device=xp.device, dtype=torch.float32)
        cell = torch.empty(b, hidden, device=xp.device, dtype=torch.float32)
        h_out = torch.empty(b, frames, hidden, device=xp.device, dtype=torch.float32)
        if frames and hidden % 4 == 0 and os.environ.get('NBASR_DENSE_MODE', 'auto') != 'f32':
            # as in the inference executor: the projection on the fp16-split GEMM, the recurrence on the fragment-ordered copy of w_hh
            # (bit-identical to the unpacked step kernel); both weights change every step, so they are packed per call
            hip.lstm_input_projection_packed(xp, frames, hip.pack_pointwise_weights(w_ih.detach().contiguous()), b_ih.detach(), b_hh.detach(), gates,
                                             hidden, hip.pointwise_workspace(b, xp.shape[1], xp.shape[2], xp.device))
            hip.lstm_recurrence_packed(gates, hip.lstm_pack_whh(w_hh.detach().contiguous()), cell, h_out)
        elif frames:
            hip.lstm_input_projection(xp, frames, w_ih.detach(), b_ih.detach(), b_hh.detach(), gates, hidden)
            hip.lstm_recurrence(gates, w_hh.detach(), cell, h_out)
        ctx.save_for_backward(xp, gates, h_out, w_ih.detach(), w_hh.detach())
        ctx.frames = frames
        return h_out

    @staticmethod
    def backward(ctx, dh):
        xp, gates, h_out, w_ih, w_hh = ctx.saved_tensors
        if ctx.frames == 0:
            return torch.zeros_like(xp[:, :, :0]), torch.zeros_like(w_ih), torch.zeros_like(w_hh), w_ih.new_zeros(w_ih.shape[0]), w_ih.new_zeros(w_ih.shape[0])
        dx, dw_ih, dw_hh, db = hip.lstm_backward(xp, ctx.frames, gates, h_out, w_ih, w_hh, dh.contiguous())
        return dx, dw_ih, dw_hh, db, db.clone()


class _PointwiseLinear(torch.autograd.Function):
    """Per-frame linear map WITHOUT activation over (B, C_in, T) -> (B, C_out, T): the CTC head (reference model.py:101-103,122-124).
    The class dimension (49) is padded to a multiple of 4 rows of zeros for the GEMMs and sliced off again."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xp, frames = _pitched(x)
        c_out = weight.shape[0]
        rows = (c_out + 3) & ~3
        w = weight.detach()
        bvec = bias.detach()
        if rows != c_out:
            w = torch.cat([w, w.new_zeros(rows - c_out, w.shape[1])])
            bvec = torch.cat([bvec, bvec.new_zeros(rows - c_out)])
        y = torch.empty(xp.shape[0], rows, xp.shape[2], device=xp.device, dtype=xp.dtype)
        hip.pointwise_linear(xp, frames, w.contiguous(), bvec.contiguous(), y)
        ctx.save_for_backward(xp, w.contiguous(), y)
        ctx.cfg = (frames, c_out)
        return y[:, :c_out, :frames]

    @staticmethod
    def backward(ctx, dy):
        xp, w, y = ctx.saved_tensors
        frames, c_out = ctx.cfg
        dyp = torch.zeros_like(y)
        dyp[:, :c_out, :frames] = dy
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dw, db = hip.dense_conv1d_backward(xp, w, y, dyp, frames, 1, need_dx, need_dw, activation=False)
        return (dx[:, :, :frames] if dx is not None else None), (dw[:c_out] if dw is not None else None), (db[:c_out] if db is not None else None)


def pointwise_linear(x, weight, bias):
    """weight (C_out, C_in) applied to every frame of (B, C_in, T), no activation, differentiable in x, weight, bias."""
    return _PointwiseLinear.apply(x, weight, bias)


def lstm(x, w_ih, w_hh, b_ih, b_hh):
    """Single-layer LSTM over (B, C, T) -> (B, T, H), differentiable in x and the four parameters (BPTT through the C ABI)."""
    return _LSTM.apply(x, w_ih, w_hh, b_ih, b_hh)


def dense_pad_conv_relu(x, weight, bias, stride=1, normalized_input=False):
    """min(relu(conv1d(zero_pad(x), weight, bias, stride)), 20) for a dense (C_out, C_in, 8) kernel, or the per-frame linear map for a
    (C_out, C_in) weight, differentiable in x, weight, bias.  ``normalized_input``: x is a LayerNorm's output (model_forward says so for
    convs 1-3), not caller data -- see the forward."""
    return _DensePadConvRelu.apply(x, weight, bias, stride, normalized_input)


def grouped_pad_conv_relu(x, weight, bias, groups, kernel, dilation):
    """min(relu(conv1d(zero_pad(x), weight, bias, dilation=dilation, groups=groups)), 20), differentiable in x, weight, bias."""
    return _GroupedPadConvRelu.apply(x, weight, bias, groups, kernel, dilation)


def layer_norm_channels(x, gamma, beta, eps=1e-3):
    """LayerNorm over the channel dimension of (B, C, T), differentiable in x, gamma, beta."""
    return _LayerNormChannels.apply(x, gamma, beta, eps)


class _SkipSum(torch.autograd.Function):
    """The node's sum (reference model.py:22, python ``sum`` over the main op's output and the flagged inputs) on the HIP skip-sum kernel,
    left to right, three tensors per launch; the gradient of a sum is the incoming gradient for every term."""

    @staticmethod
    def forward(ctx, *terms):
        pitched = [_pitched(t) for t in terms]
        frames = pitched[0][1]
        acc = None
        rest = [p for p, _ in pitched]
        while rest:
            take = ([acc] if acc is not None else []) + rest[: 3 - (acc is not None)]
            rest = rest[3 - (acc is not None):]
            out = torch.empty_like(take[0])
            hip.skip_sum(take, out, frames)
            acc = out
        ctx.n = len(terms)
        return acc[:, :, :frames]

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in range(ctx.n))


def skip_sum(terms):
    """Sum of (B, C, T) tensors, differentiable; a single term is returned as it is."""
    return terms[0] if len(terms) == 1 else _SkipSum.apply(*terms)


def model_forward(model, x):
    """The differentiable forward of an ``ASRModel`` (reference model.py:116-131 under autograd): the same layer list, every op through
    its ``torch.autograd.Function`` above, the node sums on the skip-sum kernel.  One launch group per op, nothing fused or deferred -- the
    inference executor stays the fast path; this is what ``loss.backward()`` runs through."""
    import torch.nn as nn
    from .model import SearchCell
    from .ops import PadConvRelu, Zero, Identity
    act, normalized = x, False                                 # normalized: act is a LayerNorm's output (dropout in between allowed)
    for layer in model.model:
        if isinstance(layer, PadConvRelu):
            layer.input_is_normalized = normalized             # (for this call only: a standalone call of the module gets the strict rule)
            try:
                act, normalized = layer(act), False            # routes to dense_pad_conv_relu / grouped_pad_conv_relu under autograd
            finally:
                layer.input_is_normalized = False
        elif isinstance(layer, nn.LayerNorm):
            if act.dim() == 3 and act.shape[1] == layer.normalized_shape[0]:
                act, normalized = layer_norm_channels(act, layer.weight, layer.bias, layer.eps), True
            else:
                raise RuntimeError('LayerNorm in an unexpected position of the layer list')
        elif isinstance(layer, SearchCell):
            outs = [act]
            for node in layer.nodes:
                op = node.op
                terms = [] if isinstance(op, Zero) else [op(outs[-1])]
                terms += [src for branch, src in zip(node.branch_ops, outs) if isinstance(branch, Identity)]
                outs.append(skip_sum(terms) if terms else op(outs[-1]))          # (no term at all: the `zero` op's zeros)
            act, normalized = outs[-1], False
            if layer.use_norm:
                act, normalized = layer_norm_channels(act, layer.norm_layer.weight, layer.norm_layer.bias, layer.norm_layer.eps), True
        elif isinstance(layer, nn.Dropout):
            act = layer(act)                                   # ATen's dropout (identity in eval mode or with p == 0)
        elif isinstance(layer, nn.LSTM):
            normalized = False
            act = lstm(act, layer.weight_ih_l0, layer.weight_hh_l0, layer.bias_ih_l0, layer.bias_hh_l0)      # (B, T, H)
        elif isinstance(layer, nn.Linear):
            if act.dim() == 3 and act.shape[2] == layer.in_features and act.shape[1] != layer.in_features:
                act = act.permute(0, 2, 1)                     # LSTM output (B, T, H) -> (B, H, T)
            act = pointwise_linear(act, layer.weight, layer.bias).permute(0, 2, 1)                            # (B, T, classes)
        else:
            raise TypeError(f'unsupported layer {type(layer).__name__} in the model list')
    return act
