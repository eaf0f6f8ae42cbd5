"""Storage-type generic kernels (fp32 | bf16 activations) and the kernel variants of the grouped-conv node op, through the C ABI.

* every variant of `nbasr_grouped_conv1d_node` (4 / 8 frames per lane, torch / re-laid-out weights) gives BIT-IDENTICAL results
  for a storage type: a variant is a launch geometry, never an arithmetic;
* bf16 kernels = the fp32 arithmetic on bf16-rounded inputs, rounded once: checked against the fp32 oracle evaluated on the
  same bf16-valued inputs, to one bf16 ulp (2^-8 relative) plus the fp32 tolerance.
"""
import pytest
import torch

from nb_asr_amd import hip
from oracle import asr_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BF = torch.bfloat16


def pitched(t, dtype):
    b, c, n = t.shape
    ld = hip.row_pitch(n, BF)                 # multiple of 8: the pitch every variant of either storage type accepts
    out = torch.zeros(b, c, ld, dtype=dtype, device=DEV)
    out[:, :, :n] = t.to(dtype)
    return out


def node(x, w, bias, skips, k, d, groups, dtype, variant, ln=None, on_x=False, on_s0=False, frames=None):
    frames = x.shape[2] if frames is None else frames
    xp = pitched(x, dtype)
    sp = [pitched(s, dtype) for s in skips]
    y = torch.full_like(xp, 7.0)
    wd = w.to(DEV).contiguous()
    if variant & hip.GC_WPERM:
        wd = hip.pack_grouped_weights(wd, groups)
    hip.grouped_conv1d_node(xp, wd, bias.to(DEV), sp, y, frames, groups, k, d, ln, on_x, on_s0, None, variant)
    torch.cuda.synchronize()
    assert torch.all(y[:, :, frames:] == 0)
    return y[:, :, :frames]


@pytest.mark.parametrize('dtype', [torch.float32, BF])
@pytest.mark.parametrize('cg', [6, 8, 10, 12])
@pytest.mark.parametrize('k,d', [(5, 1), (5, 2), (7, 1), (7, 2)])
def test_node_variants_are_bit_identical_and_match_the_oracle(dtype, cg, k, d):
    torch.manual_seed(cg * 100 + k * 10 + d)
    groups, b, t = 4, 3, 203
    c = cg * groups
    x = torch.randn(b, c, t).to(dtype).float()                 # bf16-representable inputs for both storage types
    skips = [torch.randn(b, c, t).to(dtype).float() for _ in range(3)]
    w = (torch.randn(c, cg, k) * 0.3).to(dtype).float()
    bias = (torch.randn(c) * 0.2).to(dtype).float()
    want = oracle.pad_conv_relu(x, w, bias, d, 1, groups) + skips[0] + skips[1] + skips[2]
    variants = (0, 1, 2, 3) + ((hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING) if dtype == torch.float32 else ())      # fp32 only
    outs = [node(x, w, bias, skips, k, d, groups, dtype, v) for v in variants]
    for v, out in zip(variants[1:], outs[1:]):
        assert torch.equal(out, outs[0]), f'variant {v} differs from variant 0'
    got = outs[0].float().cpu()
    tol = (2.0 ** -8 if dtype == BF else 0.0) * want.abs() + 2e-5 + 1e-5 * want.abs()
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())


@pytest.mark.parametrize('dtype', [torch.float32, BF])
@pytest.mark.parametrize('t', [1, 2, 7, 8, 9, 63, 64, 65, 255, 256, 257, 400, 1027])
def test_node_ragged_lengths_and_flattened_lanes(dtype, t):
    """Lanes are dealt over the flattened (utterance, chunk) index: utterance boundaries fall inside waves."""
    torch.manual_seed(t)
    groups, b, cg, k, d = 4, 5, 10, 7, 2
    c = cg * groups
    x = torch.randn(b, c, t).to(dtype).float()
    w, bias = (torch.randn(c, cg, k) * 0.3).to(dtype).float(), torch.randn(c).to(dtype).float() * 0.2
    want = oracle.pad_conv_relu(x, w, bias, d, 1, groups) + x
    for v in (0, 3) + ((hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING) if dtype == torch.float32 else ()):
        got = node(x, w, bias, [x], k, d, groups, dtype, v).float().cpu()
        tol = (2.0 ** -8 if dtype == BF else 0.0) * want.abs() + 2e-5 + 1e-5 * want.abs()
        assert bool(((got - want).abs() <= tol).all()), (v, float(((got - want).abs() / tol).max()))


@pytest.mark.parametrize('dtype', [torch.float32, BF])
def test_node_deferred_layernorm_and_epilogue_statistics(dtype):
    """LayerNorm on load (main input and skip0) and the statistics by-product, for both storage types and every variant."""
    torch.manual_seed(5)
    groups, b, t, cg, k, d = 100, 2, 77, 6, 5, 1
    c = cg * groups
    x = (torch.randn(b, c, t) * 1.5 + 0.3).to(dtype).float()
    gamma, beta = torch.rand(c) * 0.4 + 0.8, torch.randn(c) * 0.1
    w, bias = (torch.randn(c, cg, k) * 0.3).to(dtype).float(), torch.randn(c).to(dtype).float() * 0.2
    xn = oracle.layer_norm_channels(x, gamma, beta)
    want = oracle.pad_conv_relu(xn, w, bias, d, 1, groups) + xn
    xp = pitched(x, dtype)
    ld = xp.shape[2]
    stats = torch.empty(b, 2, ld, device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    ln = (stats, gamma.to(DEV), beta.to(DEV))
    results = []
    for v in tuple(range(4)) + ((hip.GC_PIPE, hip.GC_RING) if dtype == torch.float32 else ()):
        y = torch.empty_like(xp)
        ws = hip.grouped_stats_workspace(b, ld, groups, DEV)
        wd = w.to(DEV).contiguous()
        if v & hip.GC_WPERM:
            wd = hip.pack_grouped_weights(wd, groups)
        hip.grouped_conv1d_node(xp, wd, bias.to(DEV), [xp], y, t, groups, k, d, ln, True, True, ws, v)
        out_stats = torch.empty(b, 2, ld, device=DEV)
        hip.grouped_stats_finalize(ws, out_stats, c, t, groups, 1e-3)
        results.append((y[:, :, :t].clone(), out_stats[:, :, :t].clone()))
    for y, st in results[1:]:
        assert torch.equal(y, results[0][0]) and torch.allclose(st, results[0][1], rtol=1e-6, atol=1e-7)
    got = results[0][0].float().cpu()
    tol = (2.0 ** -8 if dtype == BF else 0.0) * want.abs() + 1e-4 + 1e-4 * want.abs()
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
    mean = want.mean(dim=1)
    rstd = 1.0 / torch.sqrt(want.var(dim=1, unbiased=False) + 1e-3)
    st = results[0][1].cpu()
    assert torch.allclose(st[:, 0], mean, rtol=0, atol=(3e-3 if dtype == BF else 2e-5) * float(want.abs().max()))
    assert torch.allclose(st[:, 1], rstd, rtol=2e-2 if dtype == BF else 1e-4, atol=0)


@pytest.mark.parametrize('c,t', [(600, 19), (1200, 7), (24, 37), (800, 130)])
def test_layernorm_bf16_storage(c, t):
    torch.manual_seed(c + t)
    x = (torch.randn(2, c, t) * 2 + 0.5).to(BF).float()
    gamma, beta = torch.rand(c) * 0.6 + 0.7, torch.randn(c) * 0.2
    want = oracle.layer_norm_channels(x, gamma, beta)
    xp = pitched(x, BF)
    y16, y32 = torch.full_like(xp, 3.0), torch.full(xp.shape, 3.0, device=DEV)
    hip.layernorm_channels(xp, gamma.to(DEV), beta.to(DEV), y16, t, 1e-3)
    hip.layernorm_channels(xp, gamma.to(DEV), beta.to(DEV), y32, t, 1e-3)
    assert torch.all(y16[:, :, t:] == 0) and torch.all(y32[:, :, t:] == 0)
    assert torch.allclose(y32[:, :, :t].cpu(), want, rtol=1e-5, atol=2e-5)                   # fp32 out: exact LayerNorm of the bf16 input
    assert torch.equal(y16[:, :, :t].cpu(), y32[:, :, :t].cpu().to(BF))                      # bf16 out: the same, rounded once
    stats = torch.empty(2, 2, xp.shape[2], device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    assert torch.allclose(stats[:, 0, :t].cpu(), x.mean(dim=1), atol=1e-5)
    assert torch.all(stats[:, :, t:] == 0)


@pytest.mark.parametrize('c_in,c_out,t,stride,rows', [(80, 600, 50, 1, 128), (600, 800, 300, 1, 160), (136, 200, 131, 2, 128),
                                                       (800, 1000, 257, 2, 128), (1000, 1200, 64, 2, 160), (24, 40, 37, 1, 128)])
@pytest.mark.parametrize('with_norm', [False, True])
def test_dense_conv_bf16_image_path(c_in, c_out, t, stride, rows, with_norm):
    """image (plain re-layout, or LayerNorm written as the operand) -> one-term bf16 MFMA GEMM -> bf16 rows."""
    torch.manual_seed(c_in + t)
    b = 2
    x = torch.randn(b, c_in, t).to(BF).float()
    w = (torch.randn(c_out, c_in, 8) * (2.0 / (c_in * 8)) ** 0.5).to(BF).float()
    bias = (torch.randn(c_out) * 0.2).to(BF).float()
    gamma, beta = torch.rand(c_in) * 0.4 + 0.8, torch.randn(c_in) * 0.1
    xin = oracle.layer_norm_channels(x, gamma, beta) if with_norm else x
    xin16 = xin.to(BF).float()                                   # the operand is rounded to bf16 when the image is written
    want = oracle.pad_conv_relu(xin16, w, bias, 1, stride, 1)
    xp = pitched(x, BF)
    ld = xp.shape[2]
    image = torch.empty(hip.bf16_image_bytes(b, c_in, ld), dtype=torch.uint8, device=DEV)
    stats = torch.empty(b, 2, ld, device=DEV)
    hip.bf16_image(xp, image, t, (gamma.to(DEV), beta.to(DEV)) if with_norm else None, stats if with_norm else None, 1e-3)
    t_out = (t + stride - 1) // stride
    y = torch.full((b, c_out, hip.row_pitch(t_out, BF)), 5.0, dtype=BF, device=DEV)
    packed = hip.pack_dense_weights_bf16(w.to(DEV), stride, rows)
    hip.dense_conv1d_bf16_img(image, b, c_in, t, ld, packed, c_out, 8, bias.to(DEV), y, stride, rows)
    torch.cuda.synchronize()
    assert torch.all(y[:, :, t_out:] == 0)
    got = y[:, :, :t_out].float().cpu()
    # the LayerNorm'd operand can round differently by one bf16 ulp where the fp32 value sits on a rounding boundary
    tol = 2.0 ** -8 * want.abs() + (3e-2 if with_norm else 1e-4)
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
    assert float((got - want).abs().mean()) <= 2.0 ** -9 * float(want.abs().mean()) + (2e-3 if with_norm else 1e-5)


def test_convert_and_skip_sum_and_repitch_bf16():
    torch.manual_seed(0)
    x = torch.randn(3, 24, 40, device=DEV)
    y = torch.empty(3, 24, 40, dtype=BF, device=DEV)
    hip.convert(x, y)
    assert torch.equal(y, x.to(BF))
    back = torch.empty_like(x)
    hip.convert(y, back)
    assert torch.equal(back, y.float())
    s = [torch.randn(2, 16, 24, device=DEV).to(BF) for _ in range(3)]
    out = torch.empty_like(s[0])
    hip.skip_sum(s, out, 24)
    assert torch.equal(out, (s[0].float() + s[1].float() + s[2].float()).to(BF))
    src = torch.randn(2, 5, 13, device=DEV).to(BF)
    dst = torch.full((2, 5, 16), 9.0, dtype=BF, device=DEV)
    hip.repitch(src, dst, 13)
    assert torch.equal(dst[:, :, :13], src) and torch.all(dst[:, :, 13:] == 0)


@pytest.mark.parametrize('cg,k,d', [(6, 5, 1), (12, 7, 2), (10, 5, 2), (8, 7, 1)])
def test_output_split_variant_with_deferred_layernorm(cg, k, d):
    """NBASR_GC_OSPLIT (a wave owns half of a group's output channels): bit-identical to the default kernel with LayerNorm on load
    for the main input and skip0, odd group counts (the last workgroup holds one group), three skips; no statistics flavour."""
    torch.manual_seed(cg * k + d)
    groups, b, t = 7, 3, 150
    c = cg * groups
    x = torch.randn(b, c, t) * 1.5 + 0.3
    gamma, beta = torch.rand(c) * 0.4 + 0.8, torch.randn(c) * 0.1
    w, bias = torch.randn(c, cg, k) * 0.3, torch.randn(c) * 0.2
    xp, s1, s2 = pitched(x, torch.float32), pitched(torch.randn(b, c, t), torch.float32), pitched(torch.randn(b, c, t), torch.float32)
    stats = torch.empty(b, 2, xp.shape[2], device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    ln = (stats, gamma.to(DEV), beta.to(DEV))
    outs = []
    for v in (0, hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING):
        y = torch.full_like(xp, float('nan'))
        hip.grouped_conv1d_node(xp, w.to(DEV), bias.to(DEV), [xp, s1, s2], y, t, groups, k, d, ln, True, True, None, v)
        outs.append(y)
    assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.all(outs[1][:, :, t:] == 0)
    ws = hip.grouped_stats_workspace(b, xp.shape[2], groups, DEV)
    with pytest.raises(hip.HipError, match='no statistics epilogue'):
        hip.grouped_conv1d_node(xp, w.to(DEV), bias.to(DEV), [], outs[1], t, groups, k, d, None, False, False, ws, hip.GC_OSPLIT)
    with pytest.raises(hip.HipError, match='unknown variant'):
        hip.grouped_conv1d_node(xp, w.to(DEV), bias.to(DEV), [], outs[1], t, groups, k, d, None, False, False, None, hip.GC_OSPLIT | hip.GC_FPL8)


@pytest.mark.parametrize('cg,k,d', [(6, 5, 1), (12, 7, 2), (10, 5, 2), (8, 7, 1), (12, 5, 1)])
@pytest.mark.parametrize('t', [250, 1000, 1037])
def test_ring_variant_tiles_halos_and_persistent_refills(cg, k, d, t):
    """NBASR_GC_RING (windows staged through LDS by LDS-DMA, counted waits): rows of one tile (no halo) and of
    several (halo DMAs on both sides), every
    flavour -- plain, three skips with LayerNorm on skip0, LayerNorm on load, statistics epilogue with a surplus wave in the last
    group quad -- bit-identical to the default kernel, pitch columns zero."""
    torch.manual_seed(cg * k + d + t)
    groups, b = 9, 40 if t == 250 else 6
    c = cg * groups
    x = torch.randn(b, c, t) * 1.5 + 0.3
    gamma, beta = torch.rand(c) * 0.4 + 0.8, torch.randn(c) * 0.1
    w, bias = (torch.randn(c, cg, k) * 0.3).to(DEV), (torch.randn(c) * 0.2).to(DEV)
    xp, s1, s2 = pitched(x, torch.float32), pitched(torch.randn(b, c, t), torch.float32), pitched(torch.randn(b, c, t), torch.float32)
    ld = xp.shape[2]
    stats = torch.empty(b, 2, ld, device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    ln = (stats, gamma.to(DEV), beta.to(DEV))
    flavours = {'plain': ([], None, False, False, False), 'skips': ([xp, s1, s2], ln, False, True, False), 'lnx': ([xp], ln, True, True, False),
                'stats': ([s1], None, False, False, True), 'lnx+stats': ([], ln, True, False, True)}
    for name, (skips, lnv, on_x, on_s0, with_stats) in flavours.items():
        outs = []
        for v in (0, hip.GC_RING):
            y = torch.full_like(xp, float('nan'))
            ws = hip.grouped_stats_workspace(b, ld, groups, DEV) if with_stats else None
            hip.grouped_conv1d_node(xp, w, bias, skips, y, t, groups, k, d, lnv, on_x, on_s0, ws, v)
            st = None
            if with_stats:
                st = torch.empty(b, 2, ld, device=DEV)
                hip.grouped_stats_finalize(ws, st, c, t, groups, 1e-3)
            outs.append((y, st))
        for v, (y, st) in zip(('ring',), outs[1:]):
            assert torch.equal(y, outs[0][0]), (name, v, float((y - outs[0][0]).abs().max()))
            assert torch.all(y[:, :, t:] == 0)
            if with_stats:
                assert torch.equal(st[:, :, :t], outs[0][1][:, :, :t]), (name, v, 'statistics')


@pytest.mark.parametrize('dtype', [torch.float32, BF])
@pytest.mark.parametrize('c,groups,t', [(600, 100, 1600), (1200, 100, 400), (40, 5, 801), (32, 4, 2048), (36, 6, 77)])
@pytest.mark.parametrize('kds,mask,with_ln', [(((7, 1), (7, 2), (5, 2)), 63, True), (((5, 1), (5, 1), (5, 1)), 0, True), (((5, 2), (7, 2), (7, 1)), 0b010110, False)])
def test_fused_cell_either_storage_type_equals_three_node_launches(dtype, c, groups, t, kds, mask, with_ln):
    """nbasr_grouped_cell_fused for fp32 and bf16 storage, rows up to 2048 frames (8 waves per group row, 2 groups per workgroup): the
    output equals three nbasr_grouped_conv1d_node launches BIT FOR BIT -- with bf16 storage x1 and x2 are rounded exactly where the
    node launches store them -- and the statistics by-product equals the last node launch's (per group quad: bit for bit; per pair:
    to rounding)."""
    torch.manual_seed(c + t + mask)
    b = 2
    x = (torch.randn(b, c, t) * 1.5 + 0.3).to(dtype).float()
    xp = pitched(x, dtype)
    ld = xp.shape[2]
    ln = None
    if with_ln:
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(xp, stats, t, 1e-3)
        ln = (stats, torch.rand(c, device=DEV) + 0.5, torch.randn(c, device=DEV) * 0.2)
    nodes = [((torch.randn(c, c // groups, k) * 0.3).to(dtype).float().to(DEV), (torch.randn(c) * 0.2).to(dtype).float().to(DEV), k, d) for k, d in kds]
    gpp = hip.grouped_cell_fits(c, ld, groups)
    assert gpp in (1, 2, 4)
    s = [bool(mask >> i & 1) for i in range(6)]
    x1, x2, x3 = (torch.full_like(xp, 7.0) for _ in range(3))
    ws_node = hip.grouped_stats_workspace(b, ld, groups, DEV)
    hip.grouped_conv1d_node(xp, *nodes[0][:2], [xp] if s[0] else [], x1, t, groups, *nodes[0][2:], ln, ln is not None, ln is not None and s[0], None, 0)
    hip.grouped_conv1d_node(x1, *nodes[1][:2], ([xp] if s[1] else []) + ([x1] if s[2] else []), x2, t, groups, *nodes[1][2:],
                            ln if s[1] else None, False, ln is not None and s[1], None, 0)
    hip.grouped_conv1d_node(x2, *nodes[2][:2], ([xp] if s[3] else []) + ([x1] if s[4] else []) + ([x2] if s[5] else []), x3, t, groups,
                            *nodes[2][2:], ln if s[3] else None, False, ln is not None and s[3], ws_node, 0)
    got = torch.full_like(xp, 7.0)
    ws_cell = hip.grouped_stats_workspace(b, ld, groups, DEV)
    packed = [(hip.pack_grouped_weights(w, groups), bias, k, d) for w, bias, k, d in nodes]     # [group][ci][tap][co] (ABI 4)
    hip.grouped_cell_fused(xp, packed, mask, got, t, groups, ln, ws_cell)
    assert torch.equal(got, x3), float((got.float() - x3.float()).abs().max())
    assert torch.all(got[:, :, t:] == 0)
    st_cell, st_node = torch.empty(b, 2, ld, device=DEV), torch.empty(b, 2, ld, device=DEV)
    hip.grouped_stats_finalize(ws_cell, st_cell, c, t, groups, 1e-3, gpp)
    hip.grouped_stats_finalize(ws_node, st_node, c, t, groups, 1e-3)
    if gpp == 4:
        assert torch.equal(st_cell[:, :, :t], st_node[:, :, :t])
    else:
        assert torch.allclose(st_cell[:, :, :t], st_node[:, :, :t], rtol=2e-6, atol=1e-6)


def _bf16_round(v):
    return v.to(torch.float32).to(BF).to(torch.float64)


@pytest.mark.parametrize('c,groups,t', [(600, 100, 1600), (800, 100, 1000), (1000, 100, 800), (1200, 100, 400), (40, 5, 801), (32, 4, 2048), (36, 6, 77), (48, 4, 8)])
@pytest.mark.parametrize('kds,mask,with_ln', [(((7, 1), (7, 2), (5, 2)), 63, True), (((5, 1), (5, 1), (5, 1)), 0, True), (((5, 2), (7, 2), (7, 1)), 0b010110, False),
                                              (((7, 2), (5, 1), (7, 2)), 0b101001, True)])
def test_bf16_cell_on_the_matrix_cores(c, groups, t, kds, mask, with_ln):
    """nbasr_grouped_cell_mfma (bf16 storage, v_mfma_f32_16x16x32_bf16) against a float64 restatement of what the reference computes on
    torch.bfloat16 tensors: every tensor -- the normalised cell input, x1, x2, x3 -- is rounded to bfloat16 ONCE, products and sums in
    between are exact / wide.  The kernel accumulates in fp32, so an element may land on the other side of a rounding boundary: at most
    one bf16 step off, and rarely.  Also: the pitch columns stay zero, and the vector-ALU cell (which keeps the normalised input in
    fp32) agrees to bf16 resolution."""
    torch.manual_seed(c + t + mask)
    b = 2
    x = (torch.randn(b, c, t) * 1.5 + 0.3).to(BF).float()
    xp = pitched(x, BF)
    ld = xp.shape[2]
    ln = None
    xn = x.double()
    if with_ln:
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(xp, stats, t, 1e-3)
        gamma, beta = torch.rand(c) + 0.5, torch.randn(c) * 0.2
        ln = (stats, gamma.to(DEV), beta.to(DEV))
        mean, rstd = stats[:, 0, :t].cpu().double(), stats[:, 1, :t].cpu().double()
        xn = (x.double() - mean[:, None, :]) * rstd[:, None, :] * gamma.double()[None, :, None] + beta.double()[None, :, None]
    xn = _bf16_round(xn)
    ws = [((torch.randn(c, c // groups, k) * 0.3).to(BF).float(), (torch.randn(c) * 0.2).to(BF).float(), k, d) for k, d in kds]
    s = [bool(mask >> i & 1) for i in range(6)]

    def op(v, w, bias, k, d):
        return oracle.pad_conv_relu(v, w.double(), bias.double(), d, 1, groups)
    x1 = _bf16_round(op(xn, *ws[0]) + (xn if s[0] else 0))
    x2 = _bf16_round(op(x1, *ws[1]) + (xn if s[1] else 0) + (x1 if s[2] else 0))
    want = _bf16_round(op(x2, *ws[2]) + (xn if s[3] else 0) + (x1 if s[4] else 0) + (x2 if s[5] else 0))
    assert hip.grouped_cell_mfma_fits(c, ld, groups) in (1, 2, 4)
    nodes = [(hip.grouped_cell_mfma_pack(w.to(DEV), groups), bias.to(DEV), k, d) for w, bias, k, d in ws]
    got = torch.full_like(xp, 7.0)
    hip.grouped_cell_mfma(xp, nodes, mask, got, t, groups, ln)
    assert torch.all(got[:, :, t:] == 0)
    g = got[:, :, :t].float().cpu().double()
    step = torch.clamp(want.abs() * 2.0 ** -7, min=1e-5)              # >= one bf16 step at this magnitude (floor: sign flips of a ~0 pre-activation)
    off = (g - want).abs()
    # An element of x1 / x2 that rounds the other way (fp32 vs exact accumulation next to a rounding boundary: ~1e-5 of the elements)
    # moves its consumers by weight x its bf16 step -- up to 0.125 at magnitude 16-20 -- whatever their own magnitude: so nearly every
    # element is within one step, and the few that are not are within a few hundredths (measured: 3e-5 of the elements, <= 0.05)
    assert float((off > step).double().mean()) < 2e-4, float((off > step).double().mean())
    assert float((off - 2 * step).max()) <= 0.15, float((off - 2 * step).max())
    assert float((off > 0).double().mean()) < 0.02, float((off > 0).double().mean())
    valu = torch.full_like(xp, 7.0)
    hip.grouped_cell_fused(xp, [(hip.pack_grouped_weights(w.to(DEV), groups), bias.to(DEV), k, d) for w, bias, k, d in ws], mask, valu, t, groups, ln, None)
    v = valu[:, :, :t].float().cpu().double()
    assert float((v - g).abs().max()) <= 0.05 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('c,groups,t,b', [(600, 100, 1600, 3), (800, 100, 1000, 2), (1000, 100, 800, 3), (1200, 100, 392, 5)])
def test_bf16_cell_tilings_are_bit_identical(c, groups, t, b, monkeypatch):
    """A launch of the matrix-core cell picks its tiling -- 16-frame blocks per wave (8, 10, 14 or 16; 8 or 16 for the 16-slot groups),
    groups per workgroup -- from a cost model of the sizes it is given, the batch included (what bench.py's 32 x 1600 takes is not what
    a 2-utterance test takes).  The tiling decides which wave computes an output, never the order of its sums: every tiling that fits
    must give the same bits, tail frames and pitch columns included.  NBASR_CELLM_TILING (diagnostics) forces one."""
    torch.manual_seed(c + t)
    x = (torch.randn(b, c, t) * 1.5 + 0.3).to(BF).float()
    xp = pitched(x, BF)
    ld = xp.shape[2]
    stats = torch.empty(b, 2, ld, device=DEV)
    hip.channel_stats(xp, stats, t, 1e-3)
    ln = (stats, (torch.rand(c) + 0.5).to(DEV), (torch.randn(c) * 0.2).to(DEV))
    ws = [((torch.randn(c, c // groups, k) * 0.3).to(BF).float(), (torch.randn(c) * 0.2).to(BF).float(), k, d) for k, d in ((7, 1), (5, 2), (7, 2))]
    nodes = [(hip.grouped_cell_mfma_pack(w.to(DEV), groups), bias.to(DEV), k, d) for w, bias, k, d in ws]
    monkeypatch.delenv('NBASR_CELLM_TILING', raising=False)
    want = torch.full_like(xp, 7.0)
    hip.grouped_cell_mfma(xp, nodes, 63, want, t, groups, ln)
    assert torch.all(want[:, :, t:] == 0) and torch.isfinite(want.float()).all()
    ran = 0
    for nbt in (8, 10, 14, 16):
        for gpw in (1, 2, 4):
            monkeypatch.setenv('NBASR_CELLM_TILING', f'{nbt},{gpw}')
            got = torch.full_like(xp, 7.0)
            try:
                hip.grouped_cell_mfma(xp, nodes, 63, got, t, groups, ln)
            except hip.HipError:
                continue                                   # (this tiling does not fit the row: too many waves or too much LDS)
            ran += 1
            assert torch.equal(got, want), (nbt, gpw)
    assert ran >= 3, ran


@pytest.mark.parametrize('c,groups,t', [(600, 100, 1600), (1200, 100, 1600), (800, 100, 1000)])
def test_bf16_cell_on_the_matrix_cores_contains_a_non_finite_input(c, groups, t):
    """Accepted divergence, pinned: the matrix-core cell multiplies zero WEIGHTS with real window data -- taps padded
    to an even count read the lane's tap-0 window, and with 8-channel slots one MFMA serves two column blocks half a wave tile (64 ..
    128 frames) apart through block-diagonal weights.  0 x Inf = NaN, so where the reference turns a +Inf pre-activation into 20 this
    kernel may give NaN -- but only in the group that holds the Inf, within each node's tap window of an affected frame or of that
    frame's partner in the other half of its wave tile.  Everything else is bit-identical to the same launch on a finite input, and
    no Inf ever leaves the cell (skips that are switched off are masked, not multiplied by 0)."""
    torch.manual_seed(c + t)
    b, cg, f = 2, c // groups, 700
    x = (torch.randn(b, c, t) * 1.5 + 0.3).to(BF).float()
    bad = x.clone()
    g_bad = 3
    bad[1, g_bad * cg + 2, f] = float('inf')
    ws = [((torch.rand(c, cg, k) * 0.3 + 0.01).to(BF).float(), (torch.randn(c) * 0.2).to(BF).float(), k, d) for k, d in ((7, 1), (5, 2), (7, 2))]
    nodes = [(hip.grouped_cell_mfma_pack(w.to(DEV), groups), bias.to(DEV), k, d) for w, bias, k, d in ws]
    outs = []
    for v in (x, bad):
        xp = pitched(v, BF)
        got = torch.full_like(xp, 7.0)
        hip.grouped_cell_mfma(xp, nodes, 0b110100, got, t, groups, None)       # no skip from the cell input
        outs.append(got.float().cpu())
    clean, dirty = outs
    assert torch.isfinite(clean).all()
    assert not torch.isinf(dirty).any()
    # where a NaN may surface: per node a tap window (<= 14 frames either way, 16 taken), then -- 8-channel slots only -- the same frame of
    # the partner half of its wave tile; the tiling (16-frame blocks per wave) is the launch's choice, so take the union over all of them
    near = torch.zeros(t, dtype=torch.bool)
    for nbt in (8, 10, 14, 16):
        wf, hit = 16 * nbt, {f}
        for _ in range(3):
            hit = {u for v in hit for u in range(max(0, v - 16), min(t, v + 17))}
            if cg <= 8:
                hit |= {p for p in ((v // wf) * wf + (v % wf + wf // 2) % wf for v in hit) if p < t}
        near[sorted(hit)] = True
    same = dirty == clean
    rows = slice(g_bad * cg, (g_bad + 1) * cg)
    assert same[0].all()                                                       # the other utterance
    assert same[1, :g_bad * cg].all() and same[1, (g_bad + 1) * cg:].all()     # the other groups
    assert same[1, rows, :t][:, ~near].all()                                   # this group, away from the frame
    touched = dirty[1, rows, :t][:, near]
    assert (torch.isnan(touched) | ((touched >= 0) & (touched <= 80))).all()
    assert not same[1, rows, f].all()                                          # the Inf is felt where the reference feels it


def test_bf16_cell_on_the_matrix_cores_limits():
    assert hip.grouped_cell_mfma_fits(1200, 4096, 100) == 0                  # two 32-byte-per-frame tiles of 4096 frames exceed 160 KiB
    assert hip.grouped_cell_mfma_fits(700, 1000, 100) == 0                   # 7 channels per group is not in the search space
    for shape in ((1200, 1600, 100), (600, 1600, 100), (800, 1000, 100), (1200, 256, 100)):      # (the tiling is a cost model's choice)
        assert hip.grouped_cell_mfma_fits(*shape) in (1, 2, 4)
    with pytest.raises(hip.HipError, match='not a node op'):
        hip.grouped_cell_mfma_pack(torch.randn(700, 7, 5, device=DEV), 100)


@pytest.mark.parametrize('c_in,c_out,t,b', [(600, 600, 300, 2), (1200, 1200, 77, 3), (40, 72, 1000, 1), (1000, 1000, 8, 2), (33, 130, 257, 2)])
@pytest.mark.parametrize('n_skips,with_ln', [(0, False), (1, True), (3, True), (2, False)])
def test_linear_op_on_the_bf16_matrix_cores(c_in, c_out, t, b, n_skips, with_ln):
    """nbasr_linear_fused_bf16 (round 4): the `linear` node op (ops.py:42-50 + the node's skip sum) on bf16 rows, ONE bf16 MFMA per product --
    exact bf16 x bf16 products, fp32 accumulation, skips added in fp32, one rounding.  Against a float64 evaluation of the same bf16-valued
    operands: within one bf16 rounding of the exact result (2^-8 relative) plus the fp32 accumulation noise; pitch columns zero."""
    if n_skips and c_in != c_out:
        pytest.skip('a node op maps C -> C; skips need equal shapes')
    torch.manual_seed(c_in + c_out + t + n_skips)
    x = (torch.randn(b, c_in, t) * 1.3).to(BF)
    w = (torch.randn(c_out, c_in) / c_in ** 0.5).to(BF)
    bias = (torch.randn(c_out) * 0.2).to(BF)
    skips = [(torch.randn(b, c_out, t)).to(BF) for _ in range(n_skips)]
    xp = pitched(x.float(), BF)
    ld = xp.shape[2]
    sp = [pitched(s.float(), BF) for s in skips]
    ln, xn64 = None, x.double()
    s0_64 = skips[0].double() if skips else None
    if with_ln:
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(xp, stats, t, 1e-3)
        gamma, beta = torch.rand(c_in, device=DEV) + 0.5, torch.randn(c_in, device=DEV) * 0.2
        ln = (stats, gamma, beta)
        # the pending LayerNorm of x (and of skip0, which inside a cell is the same tensor as x: use x itself as skip0 then)
        mu, var = x.double().mean(1, keepdim=True), x.double().var(1, unbiased=False, keepdim=True)
        xn64 = ((x.double() - mu) / (var + 1e-3).sqrt() * gamma.cpu().double()[None, :, None] + beta.cpu().double()[None, :, None])
        if skips:
            sp[0], s0_64 = xp, xn64                                  # skip0 = the cell input, un-normalised in storage
        xn64 = xn64.float().to(BF).double()                          # the GEMM operand is the bf16 rounding of the normalised tensor
    want = torch.einsum('oc,bct->bot', w.double(), xn64) + bias.double()[None, :, None]
    want = want.clamp(min=0.0, max=20.0)
    for i, s in enumerate(skips):
        want = want + (s0_64 if i == 0 else s.double())
    y = torch.full((b, c_out, ld), 7.0, dtype=BF, device=DEV)
    ws = hip.pointwise_bf16_workspace(b, c_in, ld, DEV)
    packed = hip.pack_pointwise_weights_bf16(w.float().to(DEV))
    hip.linear_fused_bf16(xp, t, packed, c_out, bias.float().to(DEV), sp, y, ws, ln, with_ln, with_ln and bool(skips))
    torch.cuda.synchronize()
    assert torch.all(y[:, :, t:] == 0)
    got = y[:, :, :t].float().cpu().double()
    tol = 2.0 ** -8 * want.abs() + 2e-3
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())


@pytest.mark.parametrize('c_in,hidden,t,b', [(1200, 500, 100, 3), (1200, 500, 1, 2), (72, 24, 300, 2)])
@pytest.mark.parametrize('with_ln', [False, True])
def test_lstm_projection_on_the_bf16_matrix_cores(c_in, hidden, t, b, with_ln):
    """nbasr_lstm_input_projection_bf16: gates (frames, batch, 4 H) fp32 = W_ih xn + b_ih + b_hh with xn = the bf16 rounding of the (pending)
    LayerNorm of the bf16 encoder output -- exact products, fp32 accumulation: against float64 to fp32 accumulation accuracy."""
    torch.manual_seed(c_in + hidden + t)
    x = (torch.randn(b, c_in, t) * 1.3).to(BF)
    w = (torch.randn(4 * hidden, c_in) / c_in ** 0.5).to(BF)
    b_ih, b_hh = (torch.randn(4 * hidden) * 0.2).to(BF), (torch.randn(4 * hidden) * 0.2).to(BF)
    xp = pitched(x.float(), BF)
    ld = xp.shape[2]
    ln, xn64 = None, x.double()
    if with_ln:
        stats = torch.empty(b, 2, ld, device=DEV)
        hip.channel_stats(xp, stats, t, 1e-3)
        gamma, beta = torch.rand(c_in, device=DEV) + 0.5, torch.randn(c_in, device=DEV) * 0.2
        ln = (stats, gamma, beta)
        mu, var = x.double().mean(1, keepdim=True), x.double().var(1, unbiased=False, keepdim=True)
        xn64 = ((x.double() - mu) / (var + 1e-3).sqrt() * gamma.cpu().double()[None, :, None] + beta.cpu().double()[None, :, None]).float().to(BF).double()
    want = torch.einsum('oc,bct->tbo', w.double(), xn64) + (b_ih.double() + b_hh.double())[None, None, :]
    gates = torch.full((t, b, 4 * hidden), 7.0, device=DEV)
    ws = hip.pointwise_bf16_workspace(b, c_in, ld, DEV)
    hip.lstm_input_projection_bf16(xp, t, hip.pack_pointwise_weights_bf16(w.float().to(DEV)), b_ih.float().to(DEV), b_hh.float().to(DEV), gates, hidden, ws, ln)
    torch.cuda.synchronize()
    got = gates.cpu().double()
    # with a pending LayerNorm an element of xn next to a bf16 rounding boundary may round the other way than the float64 restatement
    # (fp32 vs float64 normalisation): one bf16 step of one operand, |w| 2^-8 |xn| -- allow a few of those per output
    tol = (3e-2 if with_ln else 1e-4) * (1.0 + want.abs())
    assert bool(((got - want).abs() <= tol).all()), float(((got - want).abs() / tol).max())
