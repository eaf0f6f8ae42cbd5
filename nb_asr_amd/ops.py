"""Node operations of the search space, HIP-backed.

Interface mirror of the reference's ``nasbench_asr/model/torch/ops.py`` (same class names,
constructor arguments, parameter names and therefore ``state_dict`` keys):

* ``PadConvRelu``  (ops.py:7-30)   -> ``.conv.weight / .conv.bias``; one fused HIP launch
* ``Linear``       (ops.py:33-50)  -> ``.linear.weight / .linear.bias``; fp32-MFMA GEMM launch
* ``Identity`` / ``Zero`` (ops.py:53-68), tables ``_ops`` (ops.py:71-78), ``_branch_ops`` (80-83)

The torch modules nested inside (``nn.Conv1d``, ``nn.Linear``) are parameter containers only --
their ``forward`` is never called.  Calling an op on its own takes a ``(B, C, T)`` float32 tensor
on a HIP device; inside ``ASRModel`` the executor drives the same C entry points on pitched
workspace buffers instead (see ``executor.py``).  Dropout is the identity in eval mode or with
``p == 0``; in training mode with ``p > 0`` every op applies ATen's dropout to its output, as the
reference does (ops.py:22,29,40,48) -- with or without gradients enabled.
"""
import functools

import torch
import torch.nn as nn

from . import hip

CLAMP_MAX = 20.0
GROUPS = 100


def _pitched(x):
    """(B, C, T) tensor -> ((B, C, ld) zero-pitched tensor, T)."""
    if x.dim() != 3:
        raise ValueError(f'expected a (batch, channels, frames) tensor, got shape {tuple(x.shape)}')
    frames = x.shape[2]
    ld = hip.round_up4(frames)
    x = x.detach()
    if ld == frames and x.is_contiguous():
        return x, frames
    buf = torch.empty(x.shape[0], x.shape[1], ld, device=x.device, dtype=x.dtype)
    hip.repitch(x.contiguous(), buf, frames)
    return buf, frames


def _train_dropout(module, y):
    """The op's trailing nn.Dropout (reference ops.py:22,29 / 40,48): ATen's dropout on the op's output -- on the differentiable path
    and, round 6, on the plain one too (a training-mode module under ``torch.no_grad()``: the reference applies its masks there)."""
    if module.training and module.dropout_rate > 0:
        return torch.nn.functional.dropout(y, module.dropout_rate, True)
    return y


def _check_dropout(module):
    """The FUSED executor (fused cells, deferred LayerNorms, launch tapes, pipelined tails, captured graphs) has no dropout masks:
    a training-mode model with p > 0 runs one op at a time instead (``ASRModel.forward``); the executor-only entry points
    (``forward_async``, ``forward_graph``, ``forward_with_taps``) refuse."""
    if module.training and module.dropout_rate > 0:
        raise NotImplementedError('training-mode dropout (p > 0) is outside the fused HIP executor (forward_async / forward_graph / '
                                  'forward_with_taps); model(x) applies it op by op -- or call .eval() / build the model with dropout_rate=0.0')


class PadConvRelu(nn.Module):
    """zero-pad (look-ahead limited to ``context`` frames) -> Conv1d -> ReLU -> min(., 20)."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation, strides, groups=1, dropout_rate=0,
                 context=4, name='PadConvRelu'):
        super().__init__()
        self.name = name
        self.kernel_size, self.dilation, self.strides, self.groups = kernel_size, dilation, strides, groups
        self.dropout_rate = dropout_rate
        look_ahead = int(context / strides)
        reach = kernel_size * dilation - strides
        if look_ahead >= reach:
            self.lpad, self.rpad = 0, reach
        else:
            self.lpad, self.rpad = int((kernel_size - 1) * dilation - look_ahead), look_ahead
        # parameter container (weight (out, in/groups, k), bias (out)); never executed
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride=strides, dilation=dilation, groups=groups)

    def out_frames(self, frames):
        return (frames + self.strides - 1) // self.strides

    def forward(self, x):
        differentiable = torch.is_grad_enabled() and (x.requires_grad or self.conv.weight.requires_grad) and x.dtype == torch.float32
        if differentiable and self.groups == 1 and self.kernel_size == 8 and self.dilation == 1:
            from .autograd import dense_pad_conv_relu               # trainable on its own
            return _train_dropout(self, dense_pad_conv_relu(x, self.conv.weight, self.conv.bias, self.strides,
                                                            getattr(self, 'input_is_normalized', False)))
        if differentiable and self.groups > 1 and self.strides == 1:
            from .autograd import grouped_pad_conv_relu              # the node op is trainable on its own (first block)
            return _train_dropout(self, grouped_pad_conv_relu(x, self.conv.weight, self.conv.bias, self.groups, self.kernel_size, self.dilation))
        xp, frames = _pitched(x)
        t_out = self.out_frames(frames)
        y = torch.empty(xp.shape[0], self.conv.out_channels, hip.round_up4(t_out), device=xp.device, dtype=xp.dtype)
        if self.groups == 1:
            hip.dense_conv1d_fused(xp, frames, self.conv.weight.detach(), self.conv.bias.detach(), (), y, self.strides)
        else:
            if self.strides != 1:
                raise NotImplementedError('grouped PadConvRelu only exists with stride 1 in the search space')
            hip.grouped_conv1d_fused(xp, self.conv.weight.detach(), self.conv.bias.detach(), (), y, frames,
                                     self.groups, self.kernel_size, self.dilation)
        return _train_dropout(self, y[:, :, :t_out])


class Linear(nn.Module):
    """Per-frame fully connected layer over channels -> ReLU -> min(., 20); (B, C, T) in and out."""

    def __init__(self, in_features, out_features, dropout_rate=0, name='Linear'):
        super().__init__()
        self.name = name
        self.dropout_rate = dropout_rate
        self.linear = nn.Linear(in_features, out_features)      # parameter container

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.linear.weight.requires_grad) and x.dtype == torch.float32:
            from .autograd import dense_pad_conv_relu               # trainable on its own
            return _train_dropout(self, dense_pad_conv_relu(x, self.linear.weight, self.linear.bias, 1))
        xp, frames = _pitched(x)
        y = torch.empty(xp.shape[0], self.linear.out_features, xp.shape[2], device=xp.device, dtype=xp.dtype)
        hip.dense_conv1d_fused(xp, frames, self.linear.weight.detach().unsqueeze(-1), self.linear.bias.detach(), (), y, 1)
        return _train_dropout(self, y[:, :, :frames])


class Identity(nn.Module):
    def __init__(self, name='Identity'):
        super().__init__()
        self.name = name

    def forward(self, x):
        return x


class Zero(nn.Module):
    """An absent edge.  Contributes exact zeros (it does not propagate NaN/Inf of its input)."""

    def __init__(self, name='Zero'):
        super().__init__()
        self.name = name

    def forward(self, x):
        xp, frames = _pitched(x)
        y = torch.empty_like(xp)
        hip.skip_sum((), y, frames)
        return y[:, :, :frames]


def _grouped(kernel_size, dilation, name):
    return functools.partial(PadConvRelu, kernel_size=kernel_size, dilation=dilation, strides=1, groups=GROUPS, name=name)


# op name -> constructor taking (in_channels, out_channels, dropout_rate=...)
_ops = {
    'linear': Linear,
    'conv5': _grouped(5, 1, 'conv5'),
    'conv5d2': _grouped(5, 2, 'conv52d'),
    'conv7': _grouped(7, 1, 'conv7'),
    'conv7d2': _grouped(7, 2, 'conv52d'),      # (sic) the reference labels both dilated convs 'conv52d'
    'zero': lambda *args, **kwargs: Zero(name='zero'),
}

# skip flag -> constructor
_branch_ops = {0: Zero, 1: Identity}
