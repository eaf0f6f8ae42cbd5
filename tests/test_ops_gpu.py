"""Parity of every HIP kernel (called through the C ABI) with the golden vectors and the CPU oracle.

Sizes are small enough for the oracle to finish in seconds.  Tolerances: the kernels sum in a different
order than oneDNN, so outputs agree to a few fp32 ulps of the accumulated magnitude: rtol 2e-5 / atol 5e-6
on O(1) data (the north-star end-to-end bound is rtol 1e-4 / atol 1e-5).  Skip-sums are bit-exact.
"""
import pytest
import torch

import cases
from nb_asr_amd import hip, model as nb_model, ops as nb_ops
from oracle import asr_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RTOL, ATOL = 2e-5, 5e-6


def close(got, want, rtol=RTOL, atol=ATOL):
    got = got.detach().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(got)
    want = torch.as_tensor(want)
    assert tuple(got.shape) == tuple(want.shape), (got.shape, want.shape)
    assert torch.isfinite(got).all()
    r = cases.worst_ratio(got, want, rtol, atol)
    assert r <= 1.0, f'worst err/tol = {r:.3f}'


def pitched(x):
    """(B,C,T) cpu tensor -> zero-pitched device tensor (B,C,ld), T."""
    b, c, t = x.shape
    buf = torch.zeros(b, c, hip.round_up4(t), device=DEV)
    buf[:, :, :t] = x.to(DEV)
    return buf, t


def grouped(x, w, bias, k, d, groups, skips=()):
    xp, t = pitched(x)
    sk = [pitched(s)[0] for s in skips]
    y = torch.full_like(xp, float('nan'))
    hip.grouped_conv1d_fused(xp, w.to(DEV), bias.to(DEV), sk, y, t, groups, k, d)
    assert torch.all(y[:, :, t:] == 0)                       # pitch columns are kept at zero
    return y[:, :, :t]


@pytest.mark.parametrize('cg,k,d', cases.GCONV_CASES)
def test_grouped_conv_golden(op_fx, cg, k, d):
    c, tag = cg * 4, f'gconv/cg{cg}_k{k}_d{d}'
    p = cases.keyed_params({'conv.weight': (c, cg, k), 'conv.bias': (c,)}, tag)
    close(grouped(cases.keyed_x(tag, (2, c, 37), 2.0), p['conv.weight'], p['conv.bias'], k, d, 4), op_fx[tag])


@pytest.mark.parametrize('name,c,k,d', cases.GCONV100_CASES)
def test_grouped_conv_production_width_golden(op_fx, name, c, k, d):
    tag = f'gconv100/{name}_c{c}'
    p = cases.keyed_params({'conv.weight': (c, c // 100, k), 'conv.bias': (c,)}, tag)
    close(grouped(cases.keyed_x(tag, (1, c, 22), 2.0), p['conv.weight'], p['conv.bias'], k, d, 100), op_fx[tag])


def test_grouped_conv_clamp_golden(op_fx):
    tag = 'gconv/clamp'
    p = cases.keyed_params({'conv.weight': (24, 6, 5), 'conv.bias': (24,)}, tag)
    y = grouped(cases.keyed_x(tag, (1, 24, 16), 40.0), p['conv.weight'], p['conv.bias'], 5, 1, 4)
    assert float(y.max()) == 20.0 and float(y.min()) == 0.0
    close(y, op_fx[tag], rtol=2e-5, atol=2e-5)


def test_overflowed_preactivations_saturate_like_the_reference():
    """ops.py:27-28 (relu, then clamp_max_): a +Inf pre-activation comes out as 20, -Inf as 0, NaN as NaN.  Rounds 1-3 returned NaN for
    +-Inf (a documented divergence, VERDICT r3 missing 3); relu_clamp is now the IEEE-754-2019 maximum / minimum pair of gfx950
    (v_maximum3_f32 / v_minimum3_f32), which propagates NaN and saturates the infinities: want == got including the NaN positions.
    Every kernel family that ends in relu_clamp: the node op (default kernel and its variants), the fused cell, the dense convs (fp16
    split on the image path, exact fp32) and the `linear` op."""
    torch.manual_seed(2)
    c, groups = 24, 4
    x = torch.randn(1, c, 16)
    w = torch.randn(c, c // groups, 5) * 0.3
    bias = torch.randn(c) * 0.2
    bias[3], bias[7], bias[11] = float('inf'), float('-inf'), float('nan')
    want = oracle.pad_conv_relu(x, w, bias, 1, 1, groups)
    assert bool((want[0, 3] == 20).all()) and bool((want[0, 7] == 0).all()) and bool(want[0, 11].isnan().all())     # the reference's rule

    def same(got, ref):
        got = got.cpu()
        assert torch.equal(got.isnan(), ref.isnan())
        assert bool((got[0, 3] == 20).all()) and bool((got[0, 7] == 0).all())
        keep = ~ref.isnan()
        close(got[keep], ref[keep])
    same(grouped(x, w, bias, 5, 1, groups), want)
    xp, _ = pitched(x)
    for variant in (hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING):
        y = torch.empty_like(xp)
        hip.grouped_conv1d_node(xp, w.to(DEV), bias.to(DEV), [], y, 16, groups, 5, 1, None, False, False, None, variant)
        same(y[:, :, :16].cpu(), want)
    # the fused cell: the three nodes' biases poisoned in turn (a cell of conv5 x 3 without skips = three oracle ops in a row)
    ws = [torch.randn(c, c // groups, 5) * 0.3 for _ in range(3)]
    for node in range(3):
        bs = [torch.randn(c) * 0.2 for _ in range(3)]
        bs[node][3], bs[node][7], bs[node][11] = float('inf'), float('-inf'), float('nan')
        ref = x
        for wn, bn in zip(ws, bs):
            ref = oracle.pad_conv_relu(ref, wn, bn, 1, 1, groups)
        y = torch.empty_like(xp)
        hip.grouped_cell_fused(xp, [(hip.pack_grouped_weights(wn.to(DEV), groups), bn.to(DEV), 5, 1) for wn, bn in zip(ws, bs)], 0, y, 16, groups)
        got = y[:, :, :16].cpu()
        assert torch.equal(got.isnan(), ref.isnan()), node
        keep = ~ref.isnan()
        close(got[keep], ref[keep])
        if node == 2:
            assert bool((got[0, 3] == 20).all()) and bool((got[0, 7] == 0).all())
    # dense k = 8 conv (exact-fp32 MFMA and the packed split kernels) and the `linear` op
    wd = torch.randn(c, c, 8) * 0.1
    for stride in (1, 2):
        ref = oracle.pad_conv_relu(x, wd, bias, 1, stride, 1)
        t_out = ref.shape[2]
        y = torch.zeros(1, c, hip.round_up4(t_out), device=DEV)
        hip.dense_conv1d_fused(xp, 16, wd.to(DEV), bias.to(DEV), (), y, stride)
        same(y[:, :, :t_out].cpu(), ref)
        for scheme in ('bf16x3',):
            y = torch.zeros(1, c, hip.round_up4(t_out), device=DEV)
            hip.dense_conv1d_fused_packed(xp, 16, hip.pack_dense_weights(wd.to(DEV), stride, scheme), c, 8, bias.to(DEV), (), y, stride, None, scheme)
            same(y[:, :, :t_out].cpu(), ref)
    wl = torch.randn(c, c) * 0.2
    ref = oracle.linear_relu(x, wl, bias)
    y = torch.empty_like(xp)
    hip.linear_fused_packed(xp, 16, hip.pack_pointwise_weights(wl.to(DEV)), c, bias.to(DEV), (), y, hip.pointwise_workspace(1, c, xp.shape[2], DEV))
    same(y[:, :, :16].cpu(), ref)


@pytest.mark.parametrize('t', [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1000, 1027])
@pytest.mark.parametrize('k,d', [(5, 1), (7, 2)])
def test_grouped_conv_ragged_lengths_vs_oracle(t, k, d):
    torch.manual_seed(t * 10 + k)
    c, groups, b = 40, 4, 3
    x = torch.randn(b, c, t)
    w = torch.randn(c, c // groups, k) * 0.3
    bias = torch.randn(c) * 0.2
    skips = [torch.randn(b, c, t), torch.randn(b, c, t)]
    want = oracle.pad_conv_relu(x, w, bias, d, 1, groups) + skips[0] + skips[1]
    close(grouped(x, w, bias, k, d, groups, skips), want)


def test_grouped_conv_empty_and_errors():
    y = torch.empty(0, 24, 16, device=DEV)
    hip.grouped_conv1d_fused(torch.empty(0, 24, 16, device=DEV), torch.zeros(24, 6, 5, device=DEV), torch.zeros(24, device=DEV), (), y, 16, 4, 5, 1)
    x = torch.zeros(1, 24, 16, device=DEV)
    with pytest.raises(hip.HipError, match='unsupported'):
        hip.grouped_conv1d_fused(x, torch.zeros(24, 6, 3, device=DEV), torch.zeros(24, device=DEV), (), torch.empty_like(x), 16, 4, 3, 1)
    with pytest.raises(hip.HipError, match='multiple of 4'):
        hip.grouped_conv1d_fused(torch.zeros(1, 24, 18, device=DEV), torch.zeros(24, 6, 5, device=DEV), torch.zeros(24, device=DEV), (), torch.zeros(1, 24, 18, device=DEV), 18, 4, 5, 1)
    with pytest.raises(hip.HipError, match='HIP device'):
        hip.grouped_conv1d_fused(torch.zeros(1, 24, 16), torch.zeros(24, 6, 5, device=DEV), torch.zeros(24, device=DEV), (), torch.empty_like(x), 16, 4, 5, 1)


def dense(x, w, bias, stride, skips=()):
    b, cin, t = x.shape
    t_out = (t + stride - 1) // stride
    y = torch.full((b, w.shape[0], hip.round_up4(t_out)), float('nan'), device=DEV)
    sk = [pitched(s)[0] for s in skips]
    hip.dense_conv1d_fused(x.to(DEV).contiguous(), t, w.to(DEV), bias.to(DEV), sk, y, stride)
    assert torch.all(y[:, :, t_out:] == 0)
    return y[:, :, :t_out]


@pytest.mark.parametrize('cin,cout,t,s,b', cases.DENSE_CASES)
def test_dense_conv_golden(op_fx, cin, cout, t, s, b):
    tag = f'dense/cin{cin}_cout{cout}_t{t}_s{s}'
    p = cases.keyed_params({'conv.weight': (cout, cin, 8), 'conv.bias': (cout,)}, tag)
    close(dense(cases.keyed_x(tag, (b, cin, t)), p['conv.weight'], p['conv.bias'], s), op_fx[tag])


@pytest.mark.parametrize('cin,cout,t,s', [(8, 8, 1, 1), (8, 8, 1, 2), (12, 130, 129, 1), (12, 130, 257, 2), (80, 600, 300, 1),
                                          (600, 136, 140, 2), (20, 33, 7, 2)])
def test_dense_conv_shapes_vs_oracle(cin, cout, t, s):
    torch.manual_seed(cin + cout + t)
    x, w, bias = torch.randn(2, cin, t), torch.randn(cout, cin, 8) * (1.0 / (cin * 8) ** 0.5), torch.randn(cout) * 0.1
    close(dense(x, w, bias, s), oracle.pad_conv_relu(x, w, bias, 1, s, 1))


SCHEMES = ['bf16x3', 'f16x2']


def dense_packed(x, w, bias, stride, scheme='bf16x3'):
    b, cin, t = x.shape
    t_out = (t + stride - 1) // stride
    y = torch.full((b, w.shape[0], hip.round_up4(t_out)), float('nan'), device=DEV)
    packed = hip.pack_dense_weights(w.to(DEV), stride, scheme)
    amax = x.abs().amax(dim=(1, 2)).to(DEV) if scheme == 'f16x2' else None
    hip.dense_conv1d_fused_packed(pitched(x)[0], t, packed, w.shape[0], 8, bias.to(DEV), (), y, stride, scheme=scheme, x_absmax=amax)
    assert torch.all(y[:, :, t_out:] == 0)
    return y[:, :, :t_out]


@pytest.mark.parametrize('scheme', SCHEMES)
@pytest.mark.parametrize('cin,cout,t,s,b', cases.DENSE_CASES)
def test_dense_conv_split_golden(op_fx, cin, cout, t, s, b, scheme):
    """The operand-split paths (3-way bf16, 2-way fp16) meet the same tolerance as the fp32 MFMA path."""
    tag = f'dense/cin{cin}_cout{cout}_t{t}_s{s}'
    p = cases.keyed_params({'conv.weight': (cout, cin, 8), 'conv.bias': (cout,)}, tag)
    close(dense_packed(cases.keyed_x(tag, (b, cin, t)), p['conv.weight'], p['conv.bias'], s, scheme), op_fx[tag])


@pytest.mark.parametrize('cin,cout,t,s', [(8, 8, 1, 1), (8, 8, 1, 2), (12, 130, 129, 1), (12, 130, 257, 2), (80, 600, 300, 1),
                                          (600, 136, 140, 2), (20, 33, 7, 2), (17, 260, 515, 1), (1000, 1200, 260, 2)])
@pytest.mark.parametrize('scheme', SCHEMES)
def test_dense_conv_split_vs_fp64(cin, cout, t, s, scheme):
    """fp32-level accuracy claim: error against an fp64 evaluation is at most 2.5x that of the exact-fp32 MFMA kernel
    (both measured relative to the output scale), and absolutely below 2e-6 of the scale."""
    torch.manual_seed(cin + cout + t)
    x, w, bias = torch.randn(2, cin, t), torch.randn(cout, cin, 8) * (2.0 / (cin * 8)) ** 0.5, torch.randn(cout) * 0.1
    want = oracle.pad_conv_relu(x.double(), w.double(), bias.double(), 1, s, 1)
    scale = float(want.abs().max())
    e16 = float((dense_packed(x, w, bias, s, scheme).cpu().double() - want).pow(2).mean().sqrt()) / scale
    e32 = float((dense(x, w, bias, s).cpu().double() - want).pow(2).mean().sqrt()) / scale
    assert e16 <= max(2.5 * e32, 3e-8) and e16 < 2e-6, (e16, e32)


def test_packed_weights_extreme_values_split_exactly():
    """hi + mid + lo reproduces fp32 values of any magnitude (incl. tiny ones from the vanishing-activation regime)."""
    torch.manual_seed(0)
    cin, cout, t = 16, 128, 40
    w = torch.randn(cout, cin, 8) * torch.logspace(-20, 3, cout).view(-1, 1, 1)
    x = torch.zeros(1, cin, t)
    x[0, 3, 20] = 1.0                                     # impulse: output = bias-free copy of single weights
    y = dense_packed(x, w, torch.zeros(cout), 1)
    want = oracle.pad_conv_relu(x, w, torch.zeros(cout), 1, 1, 1)
    assert torch.equal(y.cpu(), want)                      # exact: one non-zero product per output, split is lossless


def test_f16_split_represents_values_of_any_magnitude_to_two_ulp():
    """hi + lo (two fp16 terms) carries 11 + 11 bits and a sign: a 24-bit fp32 value is represented to 2^-22 relative (2 ulp) worst
    case, whatever its magnitude -- weight ROWS spanning 23 decades (each row is normalised on its own) and inputs of any
    overall scale (each utterance is normalised on its own)."""
    torch.manual_seed(0)
    cin, cout, t = 16, 128, 40
    w = torch.randn(cout, cin, 8).sign() * (0.5 + torch.rand(cout, cin, 8)) * torch.logspace(-20, 3, cout).view(-1, 1, 1)
    for impulse in (1.0, 3.0e-9, 2.0 ** 40):
        x = torch.zeros(1, cin, t)
        x[0, 3, 20] = impulse
        y = dense_packed(x, w, torch.zeros(cout), 1, 'f16x2').cpu().double()
        want = oracle.pad_conv_relu(x.double(), w.double(), torch.zeros(cout).double(), 1, 1, 1)
        assert float(want.max()) > 0
        assert bool(((y - want).abs() <= 2.0 ** -21 * want.abs()).all()), impulse   # one product per output: 2 ulp each operand


@pytest.mark.parametrize('x_scale', [1e-9, 1.0, 1e6])
def test_f16_split_accuracy_is_scale_invariant(x_scale):
    """The vanishing-activation regime (LayerNorm outputs of 1e-9 with the reference's default init) and
    large inputs keep the same error relative to the output scale; rows of tiny weights keep their own relative accuracy."""
    torch.manual_seed(3)
    cin, cout, t, s = 136, 200, 131, 2
    x = torch.randn(2, cin, t) * x_scale
    x[1] *= 1e-3                                              # utterances are normalised separately
    # weights compensate the input scale (outputs stay below the clamp at 20, where errors would be hidden or, relative
    # to the clamped row maximum, exaggerated); rows additionally span six decades
    w = torch.randn(cout, cin, 8) * (2.0 / (cin * 8)) ** 0.5 * torch.logspace(-6, 0, cout).view(-1, 1, 1) / x_scale
    bias = torch.zeros(cout)
    want = oracle.pad_conv_relu(x.double(), w.double(), bias.double(), 1, s, 1)
    got = dense_packed(x, w, bias, s, 'f16x2').cpu().double()
    row_scale = want.abs().amax(dim=2, keepdim=True)          # per (utterance, output row)
    live = row_scale > 0
    rel = ((got - want).abs() / torch.where(live, row_scale, torch.ones_like(row_scale)))[live.expand_as(want)]
    assert float(rel.max()) < 2e-6, float(rel.max())


@pytest.mark.parametrize('b,t,h', [(3, 7, 500), (17, 5, 36), (64, 3, 500), (2, 4, 12)])
def test_lstm_recurrence_packed_is_bit_identical(b, t, h):
    """(hip.lstm_recurrence packs w_hh per call and runs the same per-frame kernel: the convenience form and the kept-packed form agree)"""
    torch.manual_seed(h + b)
    gates = torch.randn(t, b, 4 * h, device=DEV)
    w_hh = torch.randn(4 * h, h, device=DEV) * 0.2
    out0, out1 = torch.full((b, t, h), float('nan'), device=DEV), torch.full((b, t, h), float('nan'), device=DEV)
    cell = torch.empty(b, h, device=DEV)
    hip.lstm_recurrence(gates, w_hh, cell, out0)
    hip.lstm_recurrence_packed(gates, hip.lstm_pack_whh(w_hh), cell, out1)
    assert torch.isfinite(out1).all() and torch.equal(out0, out1)


def test_per_frame_recurrence_replays_its_chain_as_a_graph():
    """Round 5: nbasr_lstm_recurrence_packed replays its chain of per-frame launches as ONE cached graph per (buffers, shape, device).  The
    graph bakes in pointers, not data: new gate values through the same buffers must give the new result (against the one-launch kernel,
    which shares no launch path); more distinct buffer sets than the cache holds (16) evict without harm; a call on a stream that is being
    captured by the caller takes the plain launches inside that capture."""
    torch.manual_seed(3)
    b, t, h = 5, 9, 500
    w_hh = torch.randn(4 * h, h, device=DEV) * 0.2
    packed = hip.lstm_pack_whh(w_hh)
    ws = hip.lstm_seq_workspace(b, h, DEV)
    sets = [(torch.empty(t, b, 4 * h, device=DEV), torch.empty(b, h, device=DEV), torch.empty(b, t, h, device=DEV)) for _ in range(20)]
    for rnd in range(3):
        for gates, cell, out in sets:
            gates.copy_(torch.randn(t, b, 4 * h, device=DEV))
            out.fill_(float('nan'))
            hip.lstm_recurrence_packed(gates, packed, cell, out)
            want, cell1 = torch.full_like(out, float('nan')), torch.empty_like(cell)
            hip.lstm_recurrence_seq(gates, packed, cell1, want, ws)
            assert torch.equal(out, want) and torch.equal(cell, cell1), rnd
    gates, cell, out = sets[0]
    want = out.clone()
    out.fill_(float('nan'))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            hip.lstm_recurrence_packed(gates, packed, cell, out)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.parametrize('b,t,h', [(3, 7, 500), (17, 5, 36), (64, 3, 500), (2, 4, 12), (8, 250, 500), (33, 61, 500), (1, 1, 500), (16, 2, 512)])
def test_lstm_recurrence_in_one_launch_is_bit_identical(b, t, h):
    """All frames in one launch (w_hh resident, flag-synchronised steps) against one launch per frame: same h, bit for bit; the status
    word stays clear (no wait timed out); the final cell state is left in cell_ws as the per-frame form leaves it."""
    torch.manual_seed(h + b)
    gates = torch.randn(t, b, 4 * h, device=DEV)
    w_hh = torch.randn(4 * h, h, device=DEV) * 0.2
    packed = hip.lstm_pack_whh(w_hh)
    out0, out1 = torch.full((b, t, h), float('nan'), device=DEV), torch.full((b, t, h), float('nan'), device=DEV)
    cell0, cell1 = torch.empty(b, h, device=DEV), torch.full((b, h), float('nan'), device=DEV)
    hip.lstm_recurrence_packed(gates, packed, cell0, out0)
    ws = hip.lstm_seq_workspace(b, h, DEV)
    assert ws is not None
    hip.lstm_recurrence_seq(gates, packed, cell1, out1, ws)
    hip.lstm_seq_status(ws)
    assert torch.isfinite(out1).all() and torch.equal(out0, out1)
    assert torch.equal(cell0, cell1)


def test_lstm_recurrence_in_one_launch_under_uneven_load():
    """The hand-off of h between workgroups must not depend on placement or timing: repeat the production shape while a second stream
    keeps part of the chip busy with node kernels (so workgroups start late, run at different speeds and find warm L1s), re-using ONE
    workspace (flags and images are reset by every call), and compare every word every time."""
    torch.manual_seed(77)
    b, t, h = 64, 120, 500
    w_hh = torch.randn(4 * h, h, device=DEV) * 0.2
    packed = hip.lstm_pack_whh(w_hh)
    ws = hip.lstm_seq_workspace(b, h, DEV)
    cell = torch.empty(b, h, device=DEV)
    c, frames = 600, 1000
    xb = torch.randn(16, c, frames, device=DEV)
    yb = torch.empty_like(xb)
    wb, bb = torch.randn(c, c // 100, 5, device=DEV) * 0.1, torch.zeros(c, device=DEV)
    side = torch.cuda.Stream()
    for rep in range(6):
        gates = torch.randn(t, b, 4 * h, device=DEV)
        want = torch.empty(b, t, h, device=DEV)
        hip.lstm_recurrence_packed(gates, packed, cell, want)
        got = torch.full((b, t, h), float('nan'), device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4 + 3 * rep):
                hip.grouped_conv1d_node(xb, wb, bb, [], yb, frames, 100, 5, 1)
        hip.lstm_recurrence_seq(gates, packed, cell, got, ws)
        hip.lstm_seq_status(ws)
        torch.cuda.synchronize()
        assert torch.equal(want, got), rep


def test_lstm_recurrence_in_one_launch_limits():
    assert hip.lstm_seq_workspace(65, 500, DEV) is None          # 5 tiles x 63 slices > 256 resident workgroups
    assert hip.lstm_seq_workspace(8, 516, DEV) is None           # K does not fit one round of the 8 waves
    gates = torch.randn(2, 65, 2000, device=DEV)
    packed = hip.lstm_pack_whh(torch.randn(2000, 500, device=DEV))
    with pytest.raises(hip.HipError, match='does not apply'):
        hip.lstm_recurrence_seq(gates, packed, torch.empty(65, 500, device=DEV), torch.empty(65, 2, 500, device=DEV),
                                torch.empty(1 << 20, dtype=torch.uint8, device=DEV))


def test_layernorm_absmax_by_product():
    c, t = 40, 150
    x, xp, ln, normed = _ln_setup(c, t, seed=11)
    gamma, beta = ln[1], ln[2]
    y = torch.empty_like(xp)
    amax = torch.full((xp.shape[0],), -1.0, device=DEV)
    hip.layernorm_channels(xp, gamma, beta, y, t, 1e-3, amax)
    assert torch.equal(y, normed)
    assert torch.equal(amax.cpu(), normed.abs().amax(dim=(1, 2)).cpu())


def test_packed_scheme_mismatch_is_rejected():
    w = torch.randn(128, 16, 8, device=DEV)
    xp, t = pitched(torch.randn(1, 16, 12))
    y = torch.empty(1, 128, 12, device=DEV)
    with pytest.raises(hip.HipError):
        hip.dense_conv1d_fused_packed(xp, t, hip.pack_dense_weights(w, 1, 'f16x2'), 128, 8, torch.zeros(128, device=DEV), (), y, 1)
    with pytest.raises(hip.HipError):                       # f16x2 without the range information
        hip.dense_conv1d_fused_packed(xp, t, hip.pack_dense_weights(w, 1, 'f16x2'), 128, 8, torch.zeros(128, device=DEV), (), y, 1,
                                      scheme='f16x2')
    with pytest.raises(hip.HipError):
        hip.pack_dense_weights(w, 1, 'fp8')


@pytest.mark.parametrize('c,t,b', cases.LINEAR_CASES)
def test_linear_op_golden(op_fx, c, t, b):
    tag = f'linear/c{c}_t{t}'
    p = cases.keyed_params({'linear.weight': (c, c), 'linear.bias': (c,)}, tag)
    y = dense(cases.keyed_x(tag, (b, c, t)), p['linear.weight'].unsqueeze(-1).contiguous(), p['linear.bias'], 1)
    close(y, op_fx[tag])


def linear_packed(x, w, bias, skips=(), ln=None, on_x=False, on_s0=False, xp=None):
    """`linear` node op on the fp16 matrix cores (pre-split activations)."""
    b, c, t = x.shape
    xp = pitched(x)[0] if xp is None else xp
    y = torch.full((b, w.shape[0], xp.shape[2]), float('nan'), device=DEV)
    ws = hip.pointwise_workspace(b, c, xp.shape[2], DEV)
    hip.linear_fused_packed(xp, t, hip.pack_pointwise_weights(w.to(DEV)), w.shape[0], bias.to(DEV), list(skips), y, ws, ln, on_x, on_s0)
    assert torch.all(y[:, :, t:] == 0)
    return y[:, :, :t]


@pytest.mark.parametrize('c,t,b', cases.LINEAR_CASES)
def test_linear_op_packed_golden(op_fx, c, t, b):
    tag = f'linear/c{c}_t{t}'
    p = cases.keyed_params({'linear.weight': (c, c), 'linear.bias': (c,)}, tag)
    close(linear_packed(cases.keyed_x(tag, (b, c, t)), p['linear.weight'], p['linear.bias']), op_fx[tag])


@pytest.mark.parametrize('c_in,c_out,t,x_scale', [(24, 24, 37, 1.0), (600, 600, 300, 1.0), (136, 200, 515, 1e-9), (1200, 40, 7, 1e5),
                                                  (33 * 4, 130, 257, 1.0)])
def test_linear_op_packed_vs_fp64(c_in, c_out, t, x_scale):
    """Same accuracy bar as the split convolution: error vs fp64 within 2.5x of the exact-fp32 MFMA kernel; the per-tile and
    per-row power-of-two scalings make it independent of the overall magnitudes (frames of one utterance differ by 1e3)."""
    torch.manual_seed(c_in + t)
    x = torch.randn(2, c_in, t) * x_scale
    x[:, :, t // 2:] *= 1e-3
    w = torch.randn(c_out, c_in) * (2.0 / c_in) ** 0.5 * torch.logspace(-4, 0, c_out).view(-1, 1) / x_scale
    bias = torch.zeros(c_out)
    want = oracle.linear_relu(x.double(), w.double(), bias.double())
    scale = want.abs().amax(dim=(0, 2), keepdim=True).clamp_min(1e-300)          # per output row
    got16 = linear_packed(x, w, bias).cpu().double()
    got32 = dense(x, w.unsqueeze(-1).contiguous(), bias, 1).cpu().double()
    e16 = float(((got16 - want) / scale).pow(2).mean().sqrt())
    e32 = float(((got32 - want) / scale).pow(2).mean().sqrt())
    assert e16 <= max(2.5 * e32, 3e-8) and e16 < 2e-6, (e16, e32)


def test_linear_op_packed_skips_and_deferred_ln():
    c, t = 40, 150
    x, xp, ln, normed = _ln_setup(c, t, seed=5)
    w, bias = torch.randn(c, c) * 0.2, torch.randn(c) * 0.1
    other = pitched(torch.randn(2, c, t))[0]
    want = torch.empty_like(xp)
    hip.dense_conv1d_fused(normed, t, w.unsqueeze(-1).contiguous().to(DEV), bias.to(DEV), [normed, other], want, 1)
    got = linear_packed(x, w, bias, [xp, other], ln, True, True, xp=xp)
    close(got, want[:, :, :t].cpu(), rtol=1e-5, atol=3e-6)


@pytest.mark.parametrize('b,c_in,t,hidden', [(3, 1200, 63, 500), (2, 40, 5, 12), (16, 136, 250, 64)])
def test_lstm_input_projection_packed_vs_fp32(b, c_in, t, hidden):
    torch.manual_seed(t)
    xp, _ = pitched(torch.randn(b, c_in, t))
    w_ih, b_ih, b_hh = torch.randn(4 * hidden, c_in, device=DEV) * 0.05, torch.randn(4 * hidden, device=DEV), torch.randn(4 * hidden, device=DEV)
    g32 = torch.full((t, b, 4 * hidden), float('nan'), device=DEV)
    g16 = torch.full_like(g32, float('nan'))
    hip.lstm_input_projection(xp, t, w_ih, b_ih, b_hh, g32, hidden)
    hip.lstm_input_projection_packed(xp, t, hip.pack_pointwise_weights(w_ih), b_ih, b_hh, g16, hidden,
                                     hip.pointwise_workspace(b, c_in, xp.shape[2], DEV))
    want = torch.einsum('oc,bct->tbo', w_ih.double().cpu(), xp[:, :, :t].double().cpu()) + (b_ih + b_hh).double().cpu()
    e32 = float((g32.cpu().double() - want).abs().max())
    e16 = float((g16.cpu().double() - want).abs().max())
    assert torch.isfinite(g16).all() and e16 <= max(2.5 * e32, 1e-6), (e16, e32)


def test_zero_and_skip_sum(op_fx):
    xp, t = pitched(torch.tensor([[[float('nan'), float('inf'), 1.0, -2.0]]]))
    y = torch.full_like(xp, 7.0)
    hip.skip_sum((), y, t)
    assert torch.equal(y.cpu(), torch.from_numpy(op_fx['zero/nan']))        # exact zeros, NaN/Inf not propagated
    a, b_, c = (torch.randn(2, 24, 37) for _ in range(3))
    out = torch.empty_like(pitched(a)[0])
    hip.skip_sum([pitched(a)[0], pitched(b_)[0], pitched(c)[0]], out, 37)
    assert torch.equal(out[:, :, :37].cpu(), ((0 + a) + b_) + c)            # bit-exact, python-sum order


@pytest.mark.parametrize('op_name', cases.NODE_OPS)
def test_node_all_skip_patterns_golden(op_fx, op_name):
    c, t = 600, 12
    ins = [torch.from_numpy(cases.keyed_normal(f'node/in{i}', 3, (1, c, t))) for i in range(3)]
    p = cases.keyed_params(cases.node_shapes(op_name, c), f'node/{op_name}')
    for pattern in range(8):
        flags = [(pattern >> i) & 1 for i in range(3)]
        node = nb_model.Node(c, nb_ops._ops[op_name], [nb_ops._branch_ops[f] for f in flags]).to(DEV).eval()
        node.load_state_dict(p)
        y = node([i.to(DEV) for i in ins])
        close(y, op_fx[f'node/{op_name}_s{flags[0]}{flags[1]}{flags[2]}'])


@pytest.mark.parametrize('arch_tag', ['A', 'D', 'M'])
@pytest.mark.parametrize('use_norm', [True, False])
def test_cell_golden(op_fx, arch_tag, use_norm):
    arch = cases.ARCHS[arch_tag]
    cell = nb_model.SearchCell(600, oracle.arch_names(arch), use_norm=use_norm).to(DEV).eval()
    cell.load_state_dict(cases.keyed_params(cases.cell_shapes(arch, 600, use_norm), f'cell/{arch_tag}'))
    y = cell(cases.keyed_x(f'cell/{arch_tag}', (1, 600, 18)).to(DEV))
    close(y, op_fx[f'cell/{arch_tag}_norm{int(use_norm)}'], rtol=5e-5, atol=1e-5)


def layernorm(x, g, b, eps=1e-3):
    xp, t = pitched(x)
    y = torch.full_like(xp, float('nan'))
    hip.layernorm_channels(xp, g.to(DEV), b.to(DEV), y, t, eps)
    assert torch.all(y[:, :, t:] == 0)
    hip.layernorm_channels(xp, g.to(DEV), b.to(DEV), xp, t, eps)            # in place gives the same bits
    assert torch.equal(xp, y)
    return y[:, :, :t]


@pytest.mark.parametrize('c,t', cases.LAYERNORM_CASES)
def test_layernorm_golden(op_fx, c, t):
    tag = f'layernorm/c{c}_t{t}'
    p = cases.keyed_params({'weight': (c,), 'bias': (c,)}, tag)
    x = cases.keyed_x(tag, (2, c, t))
    x[0, :, 0] *= 1e-4
    x[1, :, 1] += 50.0
    close(layernorm(x, p['weight'], p['bias']), op_fx[tag], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('c,t', [(600, 1000), (1200, 251), (800, 1), (40, 66)])
def test_layernorm_vs_fp64_oracle(c, t):
    torch.manual_seed(c + t)
    x, g, b = torch.randn(2, c, t) * 3 + 1.5, torch.rand(c) + 0.5, torch.randn(c) * 0.1
    want = oracle.layer_norm_channels(x.double(), g.double(), b.double())
    close(layernorm(x, g, b), want, rtol=1e-5, atol=5e-6)


def lstm(x, p):
    b, t, inp = x.shape
    hid = p['weight_hh_l0'].shape[1]
    xp, _ = pitched(x.permute(0, 2, 1).contiguous())                        # encoder layout (B, C, T)
    gates = torch.empty(b * t * 4 * hid, device=DEV)
    cell = torch.empty(b * hid, device=DEV)
    h = torch.full((b, t, hid), float('nan'), device=DEV)
    hip.lstm_forward(xp, t, p['weight_ih_l0'].to(DEV), p['weight_hh_l0'].to(DEV), p['bias_ih_l0'].to(DEV),
                     p['bias_hh_l0'].to(DEV), gates, cell, h)
    return h


@pytest.mark.parametrize('inp,hid,t,b', cases.LSTM_CASES)
def test_lstm_golden(op_fx, inp, hid, t, b):
    tag = f'lstm/i{inp}_h{hid}_t{t}'
    p = cases.keyed_params({'weight_ih_l0': (4 * hid, inp), 'weight_hh_l0': (4 * hid, hid), 'bias_ih_l0': (4 * hid,),
                            'bias_hh_l0': (4 * hid,)}, tag, bias_scale=0.5)
    close(lstm(cases.keyed_x(tag, (b, t, inp)), p), op_fx[tag])


def test_lstm_model_width_vs_oracle():
    torch.manual_seed(5)
    b, t, inp, hid = 5, 30, 1200, 500
    p = {'weight_ih_l0': torch.randn(4 * hid, inp) * 0.03, 'weight_hh_l0': torch.randn(4 * hid, hid) * 0.05,
         'bias_ih_l0': torch.randn(4 * hid) * 0.1, 'bias_hh_l0': torch.randn(4 * hid) * 0.1}
    x = torch.randn(b, t, inp)
    want = oracle.lstm_forward(x, p['weight_ih_l0'], p['weight_hh_l0'], p['bias_ih_l0'], p['bias_hh_l0'])
    close(lstm(x, p), want, rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('rows,features,classes', [(1, 500, 49), (250, 500, 49), (67, 20, 49), (130, 1200, 49), (5, 8, 3)])
def test_linear_head_vs_oracle(rows, features, classes):
    torch.manual_seed(rows)
    h, w, b = torch.randn(rows, features), torch.randn(classes, features) * 0.05, torch.randn(classes)
    out = torch.full((rows, classes), float('nan'), device=DEV)
    hip.linear_head(h.to(DEV), w.to(DEV), b.to(DEV), out)
    close(out, h @ w.t() + b)
    # encoder-layout variant (use_rnn=False): x (B, features, T)
    bsz, t = 3, max(rows // 3, 1)
    x = torch.randn(bsz, features, t)
    xp, _ = pitched(x)
    out2 = torch.full((bsz, t, classes), float('nan'), device=DEV)
    hip.linear_head_bct(xp, t, w.to(DEV), b.to(DEV), out2)
    close(out2, x.permute(0, 2, 1) @ w.t() + b)


def test_standalone_op_modules_accept_unpitched_inputs():
    torch.manual_seed(1)
    m = nb_ops.PadConvRelu(24, 24, 7, 2, 1, groups=4).to(DEV).eval()
    x = torch.randn(2, 24, 37)
    want = oracle.pad_conv_relu(x, m.conv.weight.cpu(), m.conv.bias.cpu(), 2, 1, 4)
    close(m(x.to(DEV)), want.detach())
    lin = nb_ops.Linear(24, 24).to(DEV).eval()
    close(lin(x.to(DEV)), oracle.linear_relu(x, lin.linear.weight.cpu(), lin.linear.bias.cpu()).detach())
    down = nb_ops.PadConvRelu(24, 40, 8, 1, 2).to(DEV).eval()
    close(down(x.to(DEV)), oracle.pad_conv_relu(x, down.conv.weight.cpu(), down.conv.bias.cpu(), 1, 2, 1).detach())


# ---- deferred LayerNorm: one statistics pass + normalise-on-load in every consumer -----------------------------------
def _ln_setup(c, t, b=2, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(b, c, t) * 2.0 + 0.7
    g, be = torch.rand(c) + 0.5, torch.randn(c) * 0.2
    xp, _ = pitched(x)
    stats = torch.full((b, 2, xp.shape[2]), float('nan'), device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    normed = torch.empty_like(xp)
    hip.layernorm_channels(xp, g.to(DEV), be.to(DEV), normed, t, 1e-3)          # materialised reference
    return x, xp, (stats, g.to(DEV), be.to(DEV)), normed


@pytest.mark.parametrize('c,t', [(600, 37), (24, 1000), (1200, 5)])
def test_channel_stats_vs_fp64(c, t):
    x, xp, (stats, _, _), _ = _ln_setup(c, t)
    mean = x.double().mean(dim=1)
    rstd = 1.0 / torch.sqrt(x.double().var(dim=1, unbiased=False) + 1e-3)
    close(stats[:, 0, :t], mean, rtol=1e-5, atol=1e-6)
    close(stats[:, 1, :t], rstd, rtol=1e-5, atol=1e-6)
    assert torch.all(stats[:, :, t:] == 0)


@pytest.mark.parametrize('k,d', [(5, 1), (5, 2), (7, 1), (7, 2)])
@pytest.mark.parametrize('on_x,on_s0', [(True, False), (False, True), (True, True)])
def test_grouped_conv_deferred_ln_matches_materialised(k, d, on_x, on_s0):
    c, groups, t = 40, 4, 133
    x, xp, ln, normed = _ln_setup(c, t, seed=k + d)
    w, bias = torch.randn(c, c // groups, k, device=DEV) * 0.3, torch.randn(c, device=DEV) * 0.2
    other = torch.randn_like(xp)
    other[:, :, t:] = 0
    # main input: the LayerNorm-ed tensor (on_x) or an unrelated plain one; skip0: LayerNorm-ed (on_s0) or absent
    main_raw, main_ref = (xp, normed) if on_x else (other, other)
    want, got = torch.empty_like(xp), torch.full_like(xp, float('nan'))
    hip.grouped_conv1d_fused(main_ref, w, bias, [normed, other] if on_s0 else [], want, t, groups, k, d)
    hip.grouped_conv1d_fused(main_raw, w, bias, [xp, other] if on_s0 else [], got, t, groups, k, d, ln, on_x, on_s0)
    assert torch.all(got[:, :, t:] == 0)
    close(got, want.cpu(), rtol=1e-5, atol=2e-6)


def test_skip_sum_and_linear_op_deferred_ln():
    c, t = 24, 70
    x, xp, ln, normed = _ln_setup(c, t, seed=3)
    other = torch.randn_like(xp)
    other[:, :, t:] = 0
    want, got = torch.empty_like(xp), torch.full_like(xp, float('nan'))
    hip.skip_sum([normed, other], want, t)
    hip.skip_sum([xp, other], got, t, ln, True)
    close(got, want.cpu(), rtol=1e-6, atol=1e-6)
    w, bias = torch.randn(c, c, 1, device=DEV) * 0.2, torch.randn(c, device=DEV) * 0.1
    hip.dense_conv1d_fused(normed, t, w, bias, [normed], want, 1)
    hip.dense_conv1d_fused(xp, t, w, bias, [xp], got, 1, ln, True, True)
    assert torch.all(got[:, :, t:] == 0)
    close(got, want.cpu(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('stride', [1, 2])
def test_dense_conv_deferred_ln_both_paths(stride):
    c, cout, t = 40, 72, 150
    x, xp, ln, normed = _ln_setup(c, t, seed=stride)
    w, bias = torch.randn(cout, c, 8, device=DEV) * 0.1, torch.randn(cout, device=DEV) * 0.1
    t_out = (t + stride - 1) // stride
    want = torch.empty(2, cout, hip.round_up4(t_out), device=DEV)
    got32, got16 = torch.full_like(want, float('nan')), torch.full_like(want, float('nan'))
    hip.dense_conv1d_fused(normed, t, w, bias, [], want, stride)
    hip.dense_conv1d_fused(xp, t, w, bias, [], got32, stride, ln, True, False)
    hip.dense_conv1d_fused_packed(xp, t, hip.pack_dense_weights(w, stride), cout, 8, bias, [], got16, stride, ln)
    close(got32, want.cpu(), rtol=1e-5, atol=2e-6)                  # same kernel, LayerNorm applied on load: the same sums
    # the split kernel sums in another order (per channel group, then the total -- round 3): both are compared with an fp64 evaluation
    # of the same normalised input instead of with each other
    truth = torch.zeros_like(want, dtype=torch.float64, device='cpu')
    truth[:, :, :t_out] = oracle.pad_conv_relu(normed[:, :, :t].double().cpu(), w.double().cpu(), bias.double().cpu(), 1, stride, 1)
    close(want, truth, rtol=1e-5, atol=2e-6)
    close(got16, truth, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize('c,groups,t,k,d', [(40, 4, 133, 5, 1), (600, 100, 37, 7, 2), (36, 3, 260, 5, 2), (1200, 100, 250, 5, 1)])
def test_grouped_conv_epilogue_statistics(c, groups, t, k, d):
    """Statistics emitted by the convolution's epilogue == a separate statistics pass over its output."""
    torch.manual_seed(c + t)
    b = 3
    x = torch.randn(b, c, t) * 1.5
    xp, _ = pitched(x)
    w, bias = torch.randn(c, c // groups, k, device=DEV) * 0.3, torch.randn(c, device=DEV) * 0.2
    sk = torch.randn_like(xp)
    sk[:, :, t:] = 0
    y = torch.full_like(xp, float('nan'))
    got = torch.full((b, 2, xp.shape[2]), float('nan'), device=DEV)
    ws = hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV)
    hip.grouped_conv1d_fused(xp, w, bias, [sk], y, t, groups, k, d, None, False, False, got, ws, 1e-3)
    want = torch.empty_like(got)
    hip.channel_stats(y, want, t, 1e-3)
    plain = torch.empty_like(y)
    hip.grouped_conv1d_fused(xp, w, bias, [sk], plain, t, groups, k, d)
    assert torch.equal(plain, y)                                      # the output itself is unchanged
    close(got[:, 0], want[:, 0].cpu(), rtol=1e-5, atol=2e-6)
    close(got[:, 1], want[:, 1].cpu(), rtol=1e-5, atol=2e-6)
    assert torch.all(got[:, :, t:] == 0)


@pytest.mark.parametrize('c,groups,t', [(40, 4, 133), (600, 100, 37), (32, 4, 1000), (30, 5, 257), (1200, 100, 250), (42, 7, 511), (72, 12, 770), (80, 10, 1), (600, 100, 1000), (800, 100, 999)])
@pytest.mark.parametrize('kds,mask', [(((5, 1), (5, 1), (5, 1)), 0), (((7, 1), (7, 2), (5, 2)), 63), (((5, 2), (7, 2), (7, 1)), 0b101010),
                                      (((7, 2), (5, 1), (7, 2)), 0b010101)])
@pytest.mark.parametrize('with_ln', [False, True])
def test_fused_cell_is_bit_identical_to_three_node_launches(c, groups, t, kds, mask, with_ln):
    _fused_cell_vs_three_node_launches(c, groups, t, kds, mask, with_ln)


@pytest.mark.parametrize('c,groups,t', [(1200, 100, 250), (24, 2, 500), (36, 3, 1000), (32, 4, 1000), (800, 100, 499), (24, 3, 77), (48, 4, 130), (40, 5, 7)])
@pytest.mark.parametrize('kds,mask', [(((7, 1), (7, 2), (5, 2)), 63), (((5, 2), (7, 2), (7, 1)), 0b101010), (((5, 1), (5, 1), (7, 2)), 0)])
@pytest.mark.parametrize('split', ['0', '1'])
def test_fused_cell_output_channel_split_changes_no_bit(c, groups, t, kds, mask, split, monkeypatch):
    """Round 6: two waves per (group, row tile), each with half of the group's output channels (12 -> 6 + 6, 8 -> 4 + 4; the form small
    batches take by themselves): forced on and off (NBASR_CELL_OS is read at every launch), the output and the statistics are those
    of the three node launches."""
    monkeypatch.setenv('NBASR_CELL_OS', split)
    _fused_cell_vs_three_node_launches(c, groups, t, kds, mask, True)


def _fused_cell_vs_three_node_launches(c, groups, t, kds, mask, with_ln):
    torch.manual_seed(c + t + mask)
    b = 2
    x = torch.randn(b, c, t) * 1.5 + 0.3
    xp, _ = pitched(x)
    ln = None
    if with_ln:
        stats = torch.empty(b, 2, xp.shape[2], device=DEV)
        hip.channel_stats(xp, stats, t, 1e-3)
        ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
    nodes = [(torch.randn(c, c // groups, k, device=DEV) * 0.3, torch.randn(c, device=DEV) * 0.2, k, d) for k, d in kds]
    assert hip.grouped_cell_fits(c, xp.shape[2], groups)
    got = torch.full_like(xp, float('nan'))
    packed = [(hip.pack_grouped_weights(w, groups), bias, k, d) for w, bias, k, d in nodes]     # [group][ci][tap][co] (ABI 4)
    hip.grouped_cell_fused(xp, packed, mask, got, t, groups, ln)
    # reference: the per-node kernel three times
    x1, x2, x3 = (torch.full_like(xp, float('nan')) for _ in range(3))
    s = [bool(mask >> i & 1) for i in range(6)]
    hip.grouped_conv1d_fused(xp, *nodes[0][:2], [xp] if s[0] else [], x1, t, groups, *nodes[0][2:], ln, ln is not None, ln is not None and s[0])
    hip.grouped_conv1d_fused(x1, *nodes[1][:2], ([xp] if s[1] else []) + ([x1] if s[2] else []), x2, t, groups, *nodes[1][2:],
                             ln if s[1] else None, False, ln is not None and s[1])
    hip.grouped_conv1d_fused(x2, *nodes[2][:2], ([xp] if s[3] else []) + ([x1] if s[4] else []) + ([x2] if s[5] else []), x3, t, groups,
                             *nodes[2][2:], ln if s[3] else None, False, ln is not None and s[3])
    assert torch.equal(got, x3)
    assert torch.all(got[:, :, t:] == 0)
    # the statistics by-product (round 3): the same partials, hence the same (mean, rstd), as the last node launch's epilogue
    ws_cell, ws_node = hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV), hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV)
    got2, x3b = torch.full_like(xp, float('nan')), torch.full_like(xp, float('nan'))
    hip.grouped_cell_fused(xp, packed, mask, got2, t, groups, ln, ws_cell)
    hip.grouped_conv1d_node(x2, *nodes[2][:2], ([xp] if s[3] else []) + ([x1] if s[4] else []) + ([x2] if s[5] else []), x3b, t, groups,
                            *nodes[2][2:], ln if s[3] else None, False, ln is not None and s[3], ws_node, 0)
    st_cell, st_node = torch.empty(b, 2, xp.shape[2], device=DEV), torch.empty(b, 2, xp.shape[2], device=DEV)
    gpp = hip.grouped_cell_fits(c, xp.shape[2], groups)
    hip.grouped_stats_finalize(ws_cell, st_cell, c, t, groups, 1e-3, gpp)
    hip.grouped_stats_finalize(ws_node, st_node, c, t, groups, 1e-3)
    assert torch.equal(got2, x3) and torch.equal(x3b, x3)
    if gpp == 4:                                   # group quads, as the node kernel: the same partials bit for bit
        assert torch.equal(st_cell[:, :, :t], st_node[:, :, :t])
    else:                                          # group pairs / single groups (rows of more than one wave): the same statistics to rounding
        assert torch.allclose(st_cell[:, :, :t], st_node[:, :, :t], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize('c,groups,t,k', [(30, 5, 257, 7), (42, 7, 130, 5), (24, 3, 64, 7)])
def test_statistics_flavour_keeps_surplus_waves_inside_the_weights(c, groups, t, k):
    """A group count that is not a multiple of 4 leaves surplus waves in the last workgroup of the statistics flavour (they
    must reach its barrier).  Round 4 found them reading weights of groups that do not exist -- up to 3 groups past the end of
    the tensor, a memory fault when the tensor ends its device allocation.  Here weights, bias, gamma and beta are the last
    bytes of allocations of their own (>= 10 MiB requests get a segment of exactly their rounded size from the caching
    allocator), and the surplus waves must change nothing."""
    torch.manual_seed(c + t)
    b, cg = 2, c // groups
    seg = 12 << 20

    def at_end(values):
        buf = torch.empty(seg, dtype=torch.uint8, device=DEV)
        view = buf[seg - values.numel() * 4:].view(torch.float32).view(values.shape)
        view.copy_(values)
        return buf, view

    x = torch.randn(b, c, t) * 1.5 + 0.3
    xp, _ = pitched(x)
    stats = torch.empty(b, 2, xp.shape[2], device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    keep = [at_end(torch.randn(c, cg, k) * 0.3), at_end(torch.randn(c) * 0.2), at_end(torch.rand(c) + 0.5), at_end(torch.randn(c) * 0.2)]
    (w, bias, gamma, beta) = (v for _, v in keep)
    ln = (stats, gamma, beta)
    want, got = torch.full_like(xp, float('nan')), torch.full_like(xp, float('nan'))
    hip.grouped_conv1d_fused(xp, w, bias, [xp], want, t, groups, k, 1, ln, True, True)
    ws = hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV)
    hip.grouped_conv1d_node(xp, w, bias, [xp], got, t, groups, k, 1, ln, True, True, ws, 0)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # the templated kernel (here with pre-permuted weights) carries the same workgroup shape
    keep.append(at_end(hip.pack_grouped_weights(w, groups)))
    got_t, ws_t = torch.full_like(xp, float('nan')), hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV)
    hip.grouped_conv1d_node(xp, keep[-1][1], bias, [xp], got_t, t, groups, k, 1, ln, True, True, ws_t, hip.GC_WPERM)
    torch.cuda.synchronize()
    assert torch.equal(got_t, want)
    st, st_want, st_t = (torch.empty(b, 2, xp.shape[2], device=DEV) for _ in range(3))
    hip.grouped_stats_finalize(ws, st, c, t, groups, 1e-3)
    hip.grouped_stats_finalize(ws_t, st_t, c, t, groups, 1e-3)
    assert torch.equal(st_t, st)                              # the same partials (the workspace's unused tail is not compared)
    hip.channel_stats(got, st_want, t, 1e-3)
    close(st[:, :, :t], st_want[:, :, :t].cpu(), rtol=1e-5, atol=2e-6)


def test_fused_cell_limits():
    assert not hip.grouped_cell_fits(600, 2052, 100)          # > 2048 frames: more than eight 64-chunk waves per group row
    # groups per workgroup = groups per statistics partial: 1 for rows of several waves (round 4), 4 for one-wave rows
    assert hip.grouped_cell_fits(600, 1600, 100) == 1 and hip.grouped_cell_fits(1200, 1600, 100) == 1 and hip.grouped_cell_fits(1200, 2048, 100) == 1
    assert hip.grouped_cell_fits(1200, 1000, 100) == 1 and hip.grouped_cell_fits(1200, 252, 100) == 4
    assert not hip.grouped_cell_fits(700, 1000, 100)          # 7 channels per group: not a model width
    assert hip.grouped_cell_fits(800, 1000, 100) and hip.grouped_cell_fits(600, 1000, 100) and hip.grouped_cell_fits(1200, 500, 100)
    x = torch.zeros(1, 24, 16, device=DEV)
    w, bias = torch.zeros(24, 6, 3, device=DEV), torch.zeros(24, device=DEV)
    with pytest.raises(hip.HipError, match='conv5'):
        hip.grouped_cell_fused(x, [(w, bias, 3, 1)] * 3, 0, torch.empty_like(x), 16, 4)


@pytest.mark.parametrize('c,cout,t,stride,b', [(40, 72, 150, 1, 2), (40, 72, 151, 2, 2), (600, 136, 300, 1, 1), (136, 200, 515, 2, 3), (24, 40, 1, 1, 2),
                                               (1000, 136, 260, 2, 1), (72, 800, 300, 1, 1), (40, 1200, 130, 2, 2), (24, 161, 20, 1, 1)])
def test_layernorm_split_image_feeds_the_convolution(c, cout, t, stride, b):
    """LayerNorm written as the convolution's pre-split operand image + image-gathering convolution == materialised LayerNorm +
    fp16-split convolution, to fp32 resolution (the per-utterance scale comes from a bound instead of the exact maximum)."""
    torch.manual_seed(c + t)
    x = torch.randn(b, c, t) * 2.0 + 0.7
    x[0] *= 1e-6                                                # per-utterance scaling: one utterance in the eps-dominated regime
    g, be = torch.rand(c) + 0.5, torch.randn(c) * 0.2
    xp, _ = pitched(x)
    ld = xp.shape[2]
    normed = torch.empty_like(xp)
    amax = torch.empty(b, device=DEV)
    hip.layernorm_channels(xp, g.to(DEV), be.to(DEV), normed, t, 1e-3, amax)
    w, bias = torch.randn(cout, c, 8, device=DEV) * (2.0 / (c * 8)) ** 0.5, torch.randn(cout, device=DEV) * 0.1
    t_out = (t + stride - 1) // stride
    want = torch.full((b, cout, hip.round_up4(t_out)), float('nan'), device=DEV)
    got = torch.full_like(want, float('nan'))
    packed = hip.pack_dense_weights(w, stride, 'f16x2')
    hip.dense_conv1d_fused_packed(normed, t, packed, cout, 8, bias, (), want, stride, scheme='f16x2', x_absmax=amax)
    stats, bound = torch.empty(b, 2, ld, device=DEV), torch.empty(b, device=DEV)
    image = hip.split_image(b, c, ld, DEV)
    image.fill_(0x7f)                                           # poison: every row the conv reads must have been written
    hip.layernorm_split_image(xp, g.to(DEV), be.to(DEV), stats, bound, image, t, 1e-3)
    assert bool((bound >= amax).all()) and bool((bound <= 4.0 * amax + 1e-30).all())
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed, cout, 8, bias, got, stride)
    assert torch.all(got[:, :, t_out:] == 0)
    truth = oracle.pad_conv_relu(normed[:, :, :t].cpu().double(), w.cpu().double(), bias.cpu().double(), 1, stride, 1)
    scale = truth.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-300)
    e_img = float(((got[:, :, :t_out].cpu().double() - truth) / scale).abs().max())
    e_ref = float(((want[:, :, :t_out].cpu().double() - truth) / scale).abs().max())
    assert e_img <= max(2.0 * e_ref, 2e-6), (e_img, e_ref)
    # 160-row tiles: another tiling of the same sums, the K order of every output is unchanged -> bit-identical
    got160 = torch.full_like(want, float('nan'))
    packed160 = hip.pack_dense_weights(w, stride, 'f16x2', row_tile=160)
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed160, cout, 8, bias, got160, stride, row_tile=160)
    assert torch.equal(got160, got)
    with pytest.raises(hip.HipError, match='row_tile=128'):
        hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed160, cout, 8, bias, got160, stride)
    # 64-row tiles (small batches): the same again
    got64 = torch.full_like(want, float('nan'))
    packed64 = hip.pack_dense_weights(w, stride, 'f16x2', row_tile=64)
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed64, cout, 8, bias, got64, stride, row_tile=64)
    # 96-row tiles (round 4: one round of 208 workgroups for conv 3 at 16 utterances): bit-identical like the others
    got96 = torch.full_like(want, float('nan'))
    packed96 = hip.pack_dense_weights(w, stride, 'f16x2', row_tile=96)
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed96, cout, 8, bias, got96, stride, row_tile=96)
    assert torch.equal(got96, got)
    assert torch.equal(got64, got)
    # 128-frame tiles (round 5: twice the workgroups of a small batch): the same again, at every row tile
    for rows_, packed_ in ((128, packed), (160, packed160), (64, packed64), (96, packed96)):
        got128 = torch.full_like(want, float('nan'))
        hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed_, cout, 8, bias, got128, stride, row_tile=rows_, frame_tile=128)
        assert torch.equal(got128, got), rows_
    with pytest.raises(hip.HipError, match='frame_tile=64'):
        hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed, cout, 8, bias, got64, stride, frame_tile=64)


@pytest.mark.parametrize('c,cout,t,stride,b,rows', [(600, 800, 1000, 1, 2, 160), (96, 600, 301, 1, 3, 128), (112, 1000, 517, 2, 2, 128),
                                                    (64, 1200, 250, 2, 3, 160), (48, 200, 75, 2, 2, 64), (32, 161, 130, 1, 2, 96)])
def test_dense_conv_emits_the_statistics_of_its_output(c, cout, t, stride, b, rows):
    """Round 5: the image-path convolution leaves partial (mean, M2) of its output per 16 channels (stats_part), merged by
    grouped_stats_finalize with 16 as the part size: the (mean, rstd) rows that channel_stats computes in a pass of its own
    over y (the block LayerNorm behind every downsample convolution, reference model.py:92) -- and y itself is untouched."""
    torch.manual_seed(cout + t)
    x = torch.randn(b, c, t) * 1.5 + 0.2
    g, be = torch.rand(c) + 0.5, torch.randn(c) * 0.2
    xp, _ = pitched(x)
    ld = xp.shape[2]
    stats_in, bound = torch.empty(b, 2, ld, device=DEV), torch.empty(b, device=DEV)
    image = hip.split_image(b, c, ld, DEV)
    hip.layernorm_split_image(xp, g.to(DEV), be.to(DEV), stats_in, bound, image, t, 1e-3)
    w, bias = torch.randn(cout, c, 8, device=DEV) * (2.0 / (c * 8)) ** 0.5, torch.randn(cout, device=DEV) * 0.3
    bias[: cout // 3] -= 30.0                                     # a third of the channels sit at exactly 0 behind the ReLU
    t_out = (t + stride - 1) // stride
    ld_out = hip.round_up4(t_out)
    plain = torch.full((b, cout, ld_out), float('nan'), device=DEV)
    withp = torch.full_like(plain, float('nan'))
    packed = hip.pack_dense_weights(w, stride, 'f16x2', row_tile=rows)
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed, cout, 8, bias, plain, stride, row_tile=rows)
    part = torch.full((hip.dense_stats_part_floats(b, cout, ld_out),), float('nan'), device=DEV)
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed, cout, 8, bias, withp, stride, row_tile=rows, stats_part=part)
    assert torch.equal(withp, plain)
    got = torch.full((b, 2, ld_out), float('nan'), device=DEV)
    hip.grouped_stats_finalize(part, got, cout, t_out, cout, 1e-3, hip.DENSE_STATS_UNIT)
    want = torch.empty_like(got)
    hip.channel_stats(plain, want, t_out, 1e-3)
    y64 = plain[:, :, :t_out].double().cpu()
    mean64, var64 = y64.mean(dim=1), y64.var(dim=1, unbiased=False)
    rstd64 = 1.0 / (var64 + 1e-3).sqrt()
    for name, col, truth in (('mean', 0, mean64), ('rstd', 1, rstd64)):
        e_got = float((got[:, col, :t_out].double().cpu() - truth).abs().max() / truth.abs().max())
        e_ref = float((want[:, col, :t_out].double().cpu() - truth).abs().max() / truth.abs().max())
        assert e_got <= max(4.0 * e_ref, 1e-6), (name, e_got, e_ref)
    assert torch.all(got[:, :, t_out:] == 0)
    # the 16-channel unit, not the row tile, is the granule of the partials: another tiling leaves the same bits (batch invariance of
    # the statistics: the executor picks the row tile by batch size)
    other = 128 if rows != 128 else 160
    part2 = torch.full_like(part, float('nan'))
    hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, hip.pack_dense_weights(w, stride, 'f16x2', row_tile=other), cout, 8, bias,
                                          withp, stride, row_tile=other, stats_part=part2)
    assert torch.equal(withp, plain)
    live = part.view(-1, b, 2, ld_out)[:, :, :, :t_out]
    assert torch.equal(part2.view(-1, b, 2, ld_out)[:, :, :, :t_out], live) and bool(torch.isfinite(live).all())
    with pytest.raises(hip.HipError, match='too small'):
        hip.dense_conv1d_fused_packed_f16_img(image, bound, b, c, t, ld, packed, cout, 8, bias, withp, stride, row_tile=rows, stats_part=part[:-4])


@pytest.mark.parametrize('c,cout,t,b,rows', [(80, 600, 300, 4, 128), (80, 600, 1000, 2, 160), (40, 72, 7, 3, 128), (24, 161, 64, 2, 128)])
def test_first_conv_image_leg_equals_the_in_kernel_split(c, cout, t, b, rows):
    """Conv 0 on the image path (one split pass over the model input, GEMM copies operands by LDS-DMA) == conv 0 splitting the
    input in its own prologue: the same scale, the same two fp16 terms, the same K order -> bit-identical, including the
    per-utterance routing of an extreme utterance to the 3-way bf16 kernel."""
    torch.manual_seed(c * t)
    t4 = hip.round_up4(t)
    x = torch.zeros(b, c, t4)
    x[:, :, :t] = torch.randn(b, c, t) * 3.0
    x[1, :, t // 2:t] *= 2.0 ** -30                          # extreme: routed
    x[b - 1] *= 2.0 ** -35                                   # uniformly tiny: ordinary
    x = x.to(DEV)
    w, bias = torch.randn(cout, c, 8, device=DEV) * (2.0 / (c * 8)) ** 0.5, torch.randn(cout, device=DEV) * 0.1
    rng = hip.input_range(x, t, torch.empty(4 * b, device=DEV))
    p16, p16r, pb = hip.pack_dense_weights(w, 1, 'f16x2'), hip.pack_dense_weights(w, 1, 'f16x2', row_tile=rows), hip.pack_dense_weights(w, 1, 'bf16x3')
    want = torch.full((b, cout, t4), float('nan'), device=DEV)
    got = torch.full_like(want, float('nan'))
    hip.dense_conv1d_first_ranged(x, t, rng, p16, pb, cout, 8, bias, want, 1)
    image = hip.split_image(b, c, t4, DEV)
    image.fill_(0x7f)
    hip.dense_conv1d_first_ranged(x, t, rng, p16r, pb, cout, 8, bias, got, 1, image, rows)
    assert torch.equal(got, want)
    truth = oracle.pad_conv_relu(x[:, :, :t].cpu().double(), w.cpu().double(), bias.cpu().double(), 1, 1, 1)
    pre = oracle.pad_conv_relu(x[:, :, :t].cpu().double().abs(), w.cpu().double().abs(), bias.cpu().double().abs(), 1, 1, 1)
    assert float(((got[:, :, :t].cpu().double() - truth).abs() / (pre + 1e-300)).max()) <= 2e-6
    with pytest.raises(hip.HipError, match='image workspace too small'):
        hip.dense_conv1d_first_ranged(x, t, rng, p16r, pb, cout, 8, bias, got, 1, image[:16], rows)
