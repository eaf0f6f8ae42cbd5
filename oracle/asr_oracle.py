"""CPU ORACLE for the NAS-Bench-ASR acoustic-model forward pass.  TEST INFRASTRUCTURE ONLY.

Nothing in the product package (``nb_asr_amd``) imports this file: it is the checker used by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``.

It restates, as plain functions over a ``state_dict``-like mapping of tensors, the arithmetic
of the reference's PyTorch model.  The arithmetic of this path lives in ATen, so
the restatement issues THE SAME ATen calls the reference's modules make -- ``F.pad`` + ``F.conv1d``,
``F.linear``, ``F.layer_norm`` on the permuted tensor, the fused ``lstm`` op behind ``nn.LSTM`` --
in a caller-chosen dtype: float32 reproduces the reference's outputs (bit for bit for every op
but the multi-threaded reductions), bfloat16 reproduces ``model.to(torch.bfloat16)``, float64
gives a "truth" to measure both against:

* ``pad_amounts``        -- reference ``nasbench_asr/model/torch/ops.py:12-17``
* ``pad_conv_relu``      -- ``ops.py:24-30``   (zero pad -> conv1d -> relu -> min(.,20))
* ``linear_relu``        -- ``ops.py:42-50``   (per-frame x W^T + b -> relu -> min(.,20))
* ``node_forward``       -- ``model.py:13-22`` (main op on the LAST input + sum of flagged inputs,
                                                left-to-right like python ``sum``)
* ``cell_forward``       -- ``model.py:49-59`` (3 nodes, then LayerNorm over channels, eps 1e-3)
* ``layer_norm_channels``-- ``model.py:125-128`` / ``model.py:55-58``
* ``lstm_forward``       -- ``model.py:118-121`` (nn.LSTM: gates i,f,g,o; two biases; zero state)
* ``asr_forward``        -- ``model.py:116-131`` with the layer stack of ``model.py:72-103``
* ``log_softmax / output_lengths / ctc_greedy`` -- the trainer's post-logits step,
                            ``training/torch/trainer.py:217-219`` (greedy = width-1 case of its beam decode)

PINNING: the reference has no tests or golden vectors for this path.  The
oracle is pinned against outputs of the reference itself, imported in the build container:
``tests/golden/make_golden.py`` (committed) generated ``tests/golden/*.npz`` and
``tests/test_oracle_golden.py`` checks this file against them.
"""
import math

import torch
import torch.nn.functional as F

FILTERS = (600, 800, 1000, 1200)          # model.py:74
CELLS_PER_BLOCK = (3, 4, 5, 6)            # model.py:77
DOWN_KERNEL = 8                           # model.py:75
DOWN_STRIDES = (1, 1, 2, 2)               # model.py:76
FEATURES = 80                             # model.py:73
GROUPS = 100                              # ops.py:73-76
LSTM_HIDDEN = 500                         # model.py:100
LN_EPS = 1e-3                             # model.py:47,92
CLAMP = 20.0                              # ops.py:28
CONTEXT = 4                               # ops.py:8

# op name -> (kernel, dilation) for the grouped convolutions (ops.py:73-76)
CONV_OPS = {'conv5': (5, 1), 'conv5d2': (5, 2), 'conv7': (7, 1), 'conv7d2': (7, 2)}
OP_NAMES = ('linear', 'conv5', 'conv5d2', 'conv7', 'conv7d2', 'zero')   # search_space.py:6


def pad_amounts(kernel, dilation, stride, context=CONTEXT):
    """(left, right) zero padding chosen by the reference's look-ahead rule."""
    look_ahead = int(context / stride)
    span = kernel * dilation - stride
    if look_ahead >= span:
        return 0, span
    return int((kernel - 1) * dilation - look_ahead), look_ahead


def out_length(t, stride):
    """Frames after a PadConvRelu of the model: T for stride 1, ceil(T/2) for stride 2."""
    return (t + stride - 1) // stride


def pad_conv_relu(x, weight, bias, dilation, stride, groups):
    """x: (B, Cin, T) -> (B, Cout, T_out)."""
    k = weight.shape[-1]
    lpad, rpad = pad_amounts(k, dilation, stride)
    y = F.conv1d(F.pad(x, (lpad, rpad)), weight, bias, stride=stride, dilation=dilation, groups=groups)
    return torch.clamp(torch.relu(y), max=CLAMP)


def linear_relu(x, weight, bias):
    """x: (B, C, T); the same (C_out, C_in) matrix applied to every frame: permute -> nn.Linear -> relu -> clamp -> permute
    (ops.py:42-50)."""
    y = F.linear(x.permute(0, 2, 1), weight, bias)
    return torch.clamp(torch.relu(y), max=CLAMP).permute(0, 2, 1)


def layer_norm_channels(x, gamma, beta, eps=LN_EPS):
    """LayerNorm over the channel dim of (B, C, T): permute -> nn.LayerNorm(C) -> permute (model.py:55-58, 125-128)."""
    return F.layer_norm(x.permute(0, 2, 1), (x.shape[1],), gamma, beta, eps).permute(0, 2, 1)


def node_forward(inputs, op_name, flags, params, prefix):
    """inputs: list of (B,C,T) tensors (cell input, then earlier node outputs)."""
    assert len(inputs) == len(flags)
    last = inputs[-1]
    if op_name == 'zero':
        out = torch.zeros_like(last)
    elif op_name == 'linear':
        out = linear_relu(last, params[prefix + 'op.linear.weight'], params[prefix + 'op.linear.bias'])
    else:
        k, d = CONV_OPS[op_name]
        w = params[prefix + 'op.conv.weight']
        assert w.shape[-1] == k
        out = pad_conv_relu(last, w, params[prefix + 'op.conv.bias'], d, 1, GROUPS)
    for flag, src in zip(flags, inputs):
        if flag:
            out = out + src
    return out


def cell_forward(x, arch_names, params, prefix, use_norm=True):
    outs = [x]
    for j, (op_name, *flags) in enumerate(arch_names):
        outs.append(node_forward(outs, op_name, flags, params, f'{prefix}nodes.{j}.'))
    y = outs[-1]
    if use_norm:
        y = layer_norm_channels(y, params[prefix + 'norm_layer.weight'], params[prefix + 'norm_layer.bias'])
    return y


def lstm_forward(x, w_ih, w_hh, b_ih, b_hh):
    """x: (B, T, I) -> (B, T, H); single layer, zero initial state, gate order i, f, g, o: the ATen op nn.LSTM(batch_first=True)
    calls (model.py:100, 118-121)."""
    bsz, _, _ = x.shape
    hidden = w_hh.shape[1]
    if bsz == 0 or x.shape[1] == 0:
        return x.new_zeros(bsz, x.shape[1], hidden)
    zeros = x.new_zeros(1, bsz, hidden)
    out, _, _ = torch._VF.lstm(x.contiguous(), (zeros, zeros), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False, True)
    return out


def lstm_forward_loop(x, w_ih, w_hh, b_ih, b_hh):
    """The same recurrence written out frame by frame (documents what the fused op computes; tests compare the two)."""
    bsz, steps, _ = x.shape
    hidden = w_hh.shape[1]
    xg = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(bsz, hidden)
    c = x.new_zeros(bsz, hidden)
    ys = []
    for t in range(steps):
        g = xg[:, t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ys.append(h)
    return torch.stack(ys, dim=1) if ys else x.new_zeros(bsz, 0, hidden)


def arch_names(arch_vec):
    return [[OP_NAMES[node[0]]] + list(node[1:]) for node in arch_vec]


def layer_table(use_rnn=True):
    """[(index in ``model``, kind, block)] for the ModuleList of ``model.py:79-106``."""
    table, idx = [], 0
    for blk, n_cells in enumerate(CELLS_PER_BLOCK):
        table.append((idx, 'down', blk)); idx += 1
        table.append((idx, 'norm', blk)); idx += 1
        for _ in range(n_cells):
            table.append((idx, 'cell', blk)); idx += 1
    if use_rnn:
        table.append((idx, 'dropout', None)); idx += 1
        table.append((idx, 'lstm', None)); idx += 1
    table.append((idx, 'head', None))
    return table


def asr_forward(params, arch_vec, x, use_rnn=True, use_norm=True, dtype=torch.float32, taps=None, differentiable=False):
    """Full forward: x (B, 80, T) -> logits (B, ceil(ceil(T/2)/2), 49).

    ``params`` maps the reference's state_dict keys (``model.0.conv.weight`` ...) to tensors.
    ``taps``, if a dict, receives every layer's output keyed by its index in ``model``.
    ``differentiable``: keep the parameters (and x) attached, so that ATen's autograd through this op sequence yields the
    reference gradients (the tests of the backward path).
    """
    if differentiable:
        p = {k: v.to('cpu', dtype) for k, v in params.items()}
        h = x.to('cpu', dtype)
    else:
        p = {k: v.detach().to('cpu', dtype) for k, v in params.items()}
        h = x.detach().to('cpu', dtype)
    names = arch_names(arch_vec)
    for idx, kind, blk in layer_table(use_rnn):
        pre = f'model.{idx}.'
        if kind == 'down':
            h = pad_conv_relu(h, p[pre + 'conv.weight'], p[pre + 'conv.bias'], 1, DOWN_STRIDES[blk], 1)
        elif kind == 'norm':
            h = layer_norm_channels(h, p[pre + 'weight'], p[pre + 'bias'])
        elif kind == 'cell':
            h = cell_forward(h, names, p, pre, use_norm)
        elif kind == 'dropout':
            pass                                            # eval mode / p = 0
        elif kind == 'lstm':
            h = lstm_forward(h.permute(0, 2, 1), p[pre + 'weight_ih_l0'], p[pre + 'weight_hh_l0'],
                             p[pre + 'bias_ih_l0'], p[pre + 'bias_hh_l0']).permute(0, 2, 1)
        elif kind == 'head':
            h = F.linear(h.permute(0, 2, 1), p[pre + 'weight'], p[pre + 'bias'])
        if taps is not None:
            taps[idx] = h
    return h


def parameter_shapes(arch_vec, use_rnn=True, use_norm=True, num_classes=48):
    """Ordered {state_dict key: shape} for an architecture (the checkpoint compatibility surface)."""
    shapes = {}
    names = arch_names(arch_vec)
    for idx, kind, blk in layer_table(use_rnn):
        pre = f'model.{idx}.'
        if kind == 'down':
            cin = FEATURES if blk == 0 else FILTERS[blk - 1]
            shapes[pre + 'conv.weight'] = (FILTERS[blk], cin, DOWN_KERNEL)
            shapes[pre + 'conv.bias'] = (FILTERS[blk],)
        elif kind == 'norm':
            shapes[pre + 'weight'] = (FILTERS[blk],)
            shapes[pre + 'bias'] = (FILTERS[blk],)
        elif kind == 'cell':
            c = FILTERS[blk]
            for j, (op_name, *_f) in enumerate(names):
                if op_name == 'linear':
                    shapes[f'{pre}nodes.{j}.op.linear.weight'] = (c, c)
                    shapes[f'{pre}nodes.{j}.op.linear.bias'] = (c,)
                elif op_name in CONV_OPS:
                    shapes[f'{pre}nodes.{j}.op.conv.weight'] = (c, c // GROUPS, CONV_OPS[op_name][0])
                    shapes[f'{pre}nodes.{j}.op.conv.bias'] = (c,)
            if use_norm:
                shapes[pre + 'norm_layer.weight'] = (c,)
                shapes[pre + 'norm_layer.bias'] = (c,)
        elif kind == 'lstm':
            shapes[pre + 'weight_ih_l0'] = (4 * LSTM_HIDDEN, FILTERS[-1])
            shapes[pre + 'weight_hh_l0'] = (4 * LSTM_HIDDEN, LSTM_HIDDEN)
            shapes[pre + 'bias_ih_l0'] = (4 * LSTM_HIDDEN,)
            shapes[pre + 'bias_hh_l0'] = (4 * LSTM_HIDDEN,)
        elif kind == 'head':
            shapes[pre + 'weight'] = (num_classes + 1, LSTM_HIDDEN if use_rnn else FILTERS[-1])
            shapes[pre + 'bias'] = (num_classes + 1,)
    return shapes


def flops_per_forward(arch_vec, batch, frames, use_rnn=True):
    """Algorithmic FLOPs (2 per multiply-add) of one forward, by component."""
    names = arch_names(arch_vec)
    t = frames
    out = {'dense': 0.0, 'grouped': 0.0, 'linear_op': 0.0, 'lstm': 0.0, 'head': 0.0}
    cin = FEATURES
    for blk, c in enumerate(FILTERS):
        t = out_length(t, DOWN_STRIDES[blk])
        out['dense'] += 2.0 * batch * t * c * cin * DOWN_KERNEL
        for _ in range(CELLS_PER_BLOCK[blk]):
            for op_name, *_f in names:
                if op_name in CONV_OPS:
                    out['grouped'] += 2.0 * batch * t * c * (c // GROUPS) * CONV_OPS[op_name][0]
                elif op_name == 'linear':
                    out['linear_op'] += 2.0 * batch * t * c * c
        cin = c
    if use_rnn:
        out['lstm'] = 2.0 * batch * t * 4 * LSTM_HIDDEN * (FILTERS[-1] + LSTM_HIDDEN)
        out['head'] = 2.0 * batch * t * 49 * LSTM_HIDDEN
    else:
        out['head'] = 2.0 * batch * t * 49 * FILTERS[-1]
    out['total'] = math.fsum(out.values())
    return out


# ---- post-logits step (reference training/torch/trainer.py:217-219, 229-247) ------------------------------------------------
def log_softmax(logits):
    """F.log_softmax(output, dim=2) of trainer.py:218."""
    return torch.log_softmax(logits, dim=2)


def output_lengths(audio_len):
    """output_len = audio_len // 4 (trainer.py:219)."""
    return [int(n) // 4 for n in audio_len]


def ctc_greedy(logits, lengths=None, blank=0):
    """Width-1 CTC decoding: per-frame argmax (first maximum), collapse repeats, drop blanks.  Plain python loops."""
    out = []
    for b in range(logits.shape[0]):
        n = logits.shape[1] if lengths is None else max(0, min(int(lengths[b]), logits.shape[1]))
        best = torch.argmax(logits[b, :n], dim=1).tolist() if n else []
        seq, prev = [], None
        for tok in best:
            if tok != blank and tok != prev:
                seq.append(tok)
            prev = tok
        out.append(seq)
    return out
