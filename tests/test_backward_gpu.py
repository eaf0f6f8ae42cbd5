"""Backward building blocks (bottom-up): the grouped node op and the channel LayerNorm, HIP backward kernels
through the C ABI vs gradient fixtures (tests/golden/grad_fixtures.npz: ATen autograd of the reference's op sequence -- the
reference's own PadConvRelu.forward cannot be differentiated under this torch, its in-place clamp_max_ breaks ReluBackward;
LayerNorm gradients from the reference's nn.LayerNorm module)."""
import numpy as np
import pytest
import torch

import cases
from conftest import GOLDEN
from nb_asr_amd import autograd as nb_autograd, hip, ops
from nb_asr_amd.utils import keyed_normal

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def grad_fx():
    with np.load(GOLDEN / 'grad_fixtures.npz') as z:
        return {k: z[k] for k in z.files}


def close(got, want, what, rtol=1e-4):
    got, want = got.detach().double().cpu(), torch.from_numpy(want).double()
    assert tuple(got.shape) == tuple(want.shape), what
    scale = float(want.abs().max()) + 1e-30
    err = float((got - want).abs().max())
    assert err <= rtol * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def conv_case(tag, c, cg, k, shape, scale):
    p = cases.keyed_params({'conv.weight': (c, cg, k), 'conv.bias': (c,)}, tag)
    x = torch.from_numpy(keyed_normal(tag + '/x', 3, shape)) * scale
    r = torch.from_numpy(keyed_normal(tag + '/r', 5, shape))
    return x, p['conv.weight'], p['conv.bias'], r


def run_conv_grad(x, w, bias, r, groups, k, d):
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = bias.to(DEV).requires_grad_(True)
    y = nb_autograd.grouped_pad_conv_relu(xg, wg, bg, groups, k, d)
    (y * r.to(DEV)).sum().backward()
    return y, xg.grad, wg.grad, bg.grad


@pytest.mark.parametrize('cg,k,d', cases.GRAD_GCONV_CASES)
def test_grouped_node_op_gradients(grad_fx, cg, k, d):
    tag = f'grad/gconv/cg{cg}_k{k}_d{d}'
    x, w, bias, r = conv_case(tag, cg * 4, cg, k, (2, cg * 4, 37), 8.0)
    y, dx, dw, db = run_conv_grad(x, w, bias, r, 4, k, d)
    assert float(y.max()) == 20.0                                   # the clamp is active: its mask is part of what is tested
    close(dx, grad_fx[tag + '/dx'], tag + ' dx')
    close(dw, grad_fx[tag + '/dw'], tag + ' dw')
    close(db, grad_fx[tag + '/db'], tag + ' db')
    y2, dx2, dw2, db2 = run_conv_grad(x, w, bias, r, 4, k, d)          # no atomics anywhere: bit-reproducible
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize('name,c,k,d', cases.GRAD_GCONV100_CASES)
def test_grouped_node_op_gradients_production_width(grad_fx, name, c, k, d):
    tag = f'grad/gconv100/{name}_c{c}'
    x, w, bias, r = conv_case(tag, c, c // 100, k, (1, c, 70), 2.0)
    _, dx, dw, db = run_conv_grad(x, w, bias, r, 100, k, d)
    close(dx, grad_fx[tag + '/dx'], tag + ' dx')
    close(dw, grad_fx[tag + '/dw'], tag + ' dw')
    close(db, grad_fx[tag + '/db'], tag + ' db')


@pytest.mark.parametrize('t', [1, 3, 63, 64, 65, 130, 257])
def test_grouped_node_op_gradients_ragged_lengths_vs_torch_autograd(t):
    """Frame counts around the 64-frame step of the weight-gradient GEMM and the 4-frame lanes, three utterances."""
    from oracle import asr_oracle as oracle
    torch.manual_seed(t)
    c, groups, k, d, b = 40, 4, 7, 2, 3
    x, w, bias = torch.randn(b, c, t) * 3, torch.randn(c, c // groups, k) * 0.3, torch.randn(c) * 0.2
    r = torch.randn(b, c, t)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    (oracle.pad_conv_relu(xr, wr, br, d, 1, groups) * r).sum().backward()
    _, dx, dw, db = run_conv_grad(x, w, bias, r, groups, k, d)
    close(dx, xr.grad.numpy(), 'dx')
    close(dw, wr.grad.numpy(), 'dw')
    close(db, br.grad.numpy(), 'db')


@pytest.mark.parametrize('c,t', cases.GRAD_LN_CASES)
def test_layernorm_gradients(grad_fx, c, t):
    tag = f'grad/layernorm/c{c}_t{t}'
    p = cases.keyed_params({'weight': (c,), 'bias': (c,)}, tag)
    x = torch.from_numpy(keyed_normal(tag + '/x', 3, (2, c, t))).to(DEV).requires_grad_(True)
    gamma, beta = p['weight'].to(DEV).requires_grad_(True), p['bias'].to(DEV).requires_grad_(True)
    r = torch.from_numpy(keyed_normal(tag + '/r', 5, (2, c, t))).to(DEV)
    y = nb_autograd.layer_norm_channels(x, gamma, beta, 1e-3)
    (y * r).sum().backward()
    close(x.grad, grad_fx[tag + '/dx'], tag + ' dx', rtol=2e-4)
    close(gamma.grad, grad_fx[tag + '/dgamma'], tag + ' dgamma')
    close(beta.grad, grad_fx[tag + '/dbeta'], tag + ' dbeta')


def test_grouped_module_is_trainable_on_its_own():
    """ops.PadConvRelu with groups > 1 routes through the autograd function when a gradient is required: one SGD step on
    its own parameters lowers a loss; without grad the ordinary fused launch runs."""
    torch.manual_seed(0)
    m = ops._ops['conv5'](600, 600).to(DEV)
    x = torch.randn(2, 600, 50, device=DEV)
    target = torch.rand(2, 600, 50, device=DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    losses = []
    for _ in range(5):
        opt.zero_grad()
        loss = ((m(x) - target) ** 2).mean()
        loss.backward()
        assert m.conv.weight.grad is not None and torch.isfinite(m.conv.weight.grad).all()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]
    with torch.no_grad():
        assert m(x).grad_fn is None


def test_backward_argument_errors():
    lib = hip.load_library()
    rc = lib.nbasr_grouped_conv1d_backward(16, 16, 16, 16, 16, 16, None, 16, 1, 600, 16, 16, 100, 5, 1, None)
    assert rc == -3 and b'come together' in lib.nbasr_last_error()
    rc = lib.nbasr_grouped_conv1d_backward(16, 16, 16, 16, 16, None, None, None, 1, 600, 10, 10, 100, 5, 1, None)
    assert rc == -2
    assert lib.nbasr_grouped_conv1d_backward_workspace_bytes(64, 1200, 100, 5) == 100 * 64 * 4 * 256 * 4


# ---- dense downsample convs and the per-frame linear op (GEMM-shaped backward on the exact-fp32 MFMA GEMMs) --------------------
def dense_case(tag, w_shape, x_shape, scale):
    names = {'conv.weight': w_shape, 'conv.bias': (w_shape[0],)} if len(w_shape) == 3 else {'linear.weight': w_shape, 'linear.bias': (w_shape[0],)}
    p = cases.keyed_params(names, tag)
    w, bias = [p[k] for k in names]
    x = torch.from_numpy(keyed_normal(tag + '/x', 3, x_shape)) * scale
    return x, w, bias


def run_dense_grad(x, w, bias, r, stride):
    xg, wg, bg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    y = nb_autograd.dense_pad_conv_relu(xg, wg, bg, stride)
    (y * r.to(DEV)).sum().backward()
    return y, xg.grad, wg.grad, bg.grad


@pytest.mark.parametrize('c_in,c_out,stride,b,t', cases.GRAD_DENSE_CASES)
def test_dense_conv_gradients(grad_fx, c_in, c_out, stride, b, t):
    tag = f'grad/dense/c{c_in}_{c_out}_s{stride}_t{t}'
    x, w, bias = dense_case(tag, (c_out, c_in, 8), (b, c_in, t), 6.0)
    t_out = (t + stride - 1) // stride
    r = torch.from_numpy(keyed_normal(tag + '/r', 5, (b, c_out, t_out)))
    y, dx, dw, db = run_dense_grad(x, w, bias, r, stride)
    assert float(y.max()) == 20.0 and tuple(dx.shape) == (b, c_in, t)
    close(dx, grad_fx[tag + '/dx'], tag + ' dx')
    close(dw, grad_fx[tag + '/dw'], tag + ' dw')
    close(db, grad_fx[tag + '/db'], tag + ' db')
    y2, dx2, dw2, db2 = run_dense_grad(x, w, bias, r, stride)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)      # no atomics: bit-reproducible


@pytest.mark.parametrize('c_in,c_out,b,t', cases.GRAD_LINEAR_CASES)
def test_linear_op_gradients(grad_fx, c_in, c_out, b, t):
    tag = f'grad/linear/c{c_in}_{c_out}_t{t}'
    x, w, bias = dense_case(tag, (c_out, c_in), (b, c_in, t), 6.0)
    r = torch.from_numpy(keyed_normal(tag + '/r', 5, (b, c_out, t)))
    y, dx, dw, db = run_dense_grad(x, w, bias, r, 1)
    assert float(y.max()) == 20.0
    close(dx, grad_fx[tag + '/dx'], tag + ' dx')
    close(dw, grad_fx[tag + '/dw'], tag + ' dw')
    close(db, grad_fx[tag + '/db'], tag + ' db')


@pytest.mark.parametrize('c_in,c_out,stride,b,t', [(600, 136, 1, 2, 131), (136, 200, 2, 3, 64), (80, 600, 1, 1, 5), (1000, 24, 2, 1, 1)])
def test_dense_conv_gradients_vs_torch_autograd(c_in, c_out, stride, b, t):
    """Production-width channels, ragged lengths and lengths shorter than the kernel, against ATen's autograd of the oracle."""
    from oracle import asr_oracle as oracle
    torch.manual_seed(c_in + t)
    x = torch.randn(b, c_in, t) * 3.0
    w = torch.randn(c_out, c_in, 8) * (2.0 / (c_in * 8)) ** 0.5
    bias = torch.randn(c_out) * 0.2
    t_out = (t + stride - 1) // stride
    r = torch.randn(b, c_out, t_out)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    (oracle.pad_conv_relu(xr, wr, br, 1, stride, 1) * r).sum().backward()
    y, dx, dw, db = run_dense_grad(x, w, bias, r, stride)
    close(dx, xr.grad.numpy(), 'dx', rtol=2e-4)
    close(dw, wr.grad.numpy(), 'dw', rtol=2e-4)
    close(db, br.grad.numpy(), 'db', rtol=2e-4)


@pytest.mark.parametrize('c_in,stride,b,t', [(6, 1, 2, 1), (40, 1, 3, 70), (34, 2, 2, 65), (33, 2, 1, 64), (70, 1, 1, 33)])
def test_conv_fold_adds_the_tap_rows(c_in, stride, b, t):
    """nbasr_conv_fold against a loop over taps: cols (T', B, C_in * 8) time-major -> dx (B, C_in, ld), pitch columns zero."""
    import ctypes
    t_out = (t + stride - 1) // stride
    lpad = hip.pad_amounts(8, 1, stride)[0]
    torch.manual_seed(c_in * 7 + t)
    cols = torch.randn(t_out, b, c_in * 8)
    ld = hip.round_up4(t) + 4
    dx = torch.full((b, c_in, ld), 7.0, device=DEV)
    lib = hip.load_library()
    rc = lib.nbasr_conv_fold(cols.to(DEV).data_ptr(), dx.data_ptr(), b, c_in, t, ld, t_out, 8, stride, lpad, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.nbasr_last_error()
    want = torch.zeros(b, c_in, ld, dtype=torch.float64)
    c4 = cols.double().view(t_out, b, c_in, 8)
    for j in range(8):
        for to in range(t_out):
            u = to * stride + j - lpad
            if 0 <= u < t:
                want[:, :, u] += c4[to, :, :, j]
    got = dx.cpu().double()
    assert torch.equal(got[:, :, t:], torch.zeros_like(got[:, :, t:]))
    assert float((got - want).abs().max()) <= 1e-5
    for bad in ((7, 1), (8, 3)):
        assert lib.nbasr_conv_fold(cols.to(DEV).data_ptr(), dx.data_ptr(), b, c_in, t, ld, t_out, bad[0], bad[1], lpad, 0) != 0


@pytest.mark.parametrize('c_in,c_out,stride,b,t', [(600, 800, 1, 2, 97), (800, 1000, 2, 2, 131), (24, 40, 2, 3, 5)])
def test_dense_input_gradient_on_the_split_gemm_matches_the_exact_route(monkeypatch, c_in, c_out, stride, b, t):
    """Default route of the k = 8 input gradient (one fp16-split GEMM + nbasr_conv_fold) against NBASR_DENSE_MODE=f32 (zero-stuffed
    exact-fp32 conv): same numbers to the fp32-accurate split's error, at the benchmark model's widths and ragged lengths."""
    torch.manual_seed(c_in + stride)
    t_out = (t + stride - 1) // stride
    ld_in, ld_out = hip.round_up4(t), hip.round_up4(t_out)
    x = torch.zeros(b, c_in, ld_in, device=DEV)
    x[:, :, :t] = torch.randn(b, c_in, t, device=DEV) * 3.0
    w = (torch.randn(c_out, c_in, 8) * (2.0 / (c_in * 8)) ** 0.5).to(DEV)
    y = torch.zeros(b, c_out, ld_out, device=DEV)                     # one saved output for both routes: the same ReLU / clamp mask
    y[:, :, :t_out] = (torch.randn(b, c_out, t_out, device=DEV) * 8.0).clamp(0.0, 20.0)
    dy = torch.zeros(b, c_out, ld_out, device=DEV)
    dy[:, :, :t_out] = torch.randn(b, c_out, t_out, device=DEV) * 1e-3
    dx, dw, db = hip.dense_conv1d_backward(x, w, y, dy, t, stride)
    monkeypatch.setenv('NBASR_DENSE_MODE', 'f32')
    dx_e, dw_e, db_e = hip.dense_conv1d_backward(x, w, y, dy, t, stride)
    assert not torch.equal(dx, dx_e)                                  # (two routes indeed)
    assert torch.equal(dx[:, :, t:], torch.zeros_like(dx[:, :, t:]))
    scale = float(dx_e.abs().max())
    assert float((dx - dx_e).abs().max()) <= 2e-5 * scale, (float((dx - dx_e).abs().max()), scale)
    assert float((dw - dw_e).abs().max()) <= 2e-5 * float(dw_e.abs().max())
    assert float((db - db_e).abs().max()) <= 2e-5 * float(db_e.abs().max())


def test_dense_ops_train_through_the_modules():
    """ops.PadConvRelu (dense) and ops.Linear route to the differentiable functions when a gradient is required: an SGD step
    on each lowers a loss."""
    torch.manual_seed(0)
    for mod, x in ((ops.PadConvRelu(24, 40, 8, 1, 2).to(DEV), torch.randn(2, 24, 50, device=DEV)),
                   (ops.Linear(24, 24).to(DEV), torch.randn(2, 24, 50, device=DEV))):
        target = torch.rand(2, mod(x).shape[1], mod(x).shape[2], device=DEV)
        losses = []
        for _ in range(4):
            loss = ((mod(x) - target) ** 2).mean()
            losses.append(float(loss))
            mod.zero_grad()
            loss.backward()
            with torch.no_grad():
                for p in mod.parameters():
                    p -= 0.05 * p.grad
        assert losses[-1] < losses[0], losses


def test_lstm_bptt_routes_agree(monkeypatch):
    """The LSTM's backward GEMMs on the fp16-split kernel (default) and on the exact-fp32 one (NBASR_DENSE_MODE=f32): same gradients to
    the split's error, from the same saved forward."""
    torch.manual_seed(3)
    c, hidden, b, t = 72, 36, 5, 23
    params = [(torch.randn(4 * hidden, c) * 0.2), (torch.randn(4 * hidden, hidden) * 0.2), torch.randn(4 * hidden) * 0.1, torch.randn(4 * hidden) * 0.1]
    x, r = torch.randn(b, c, t), torch.randn(b, t, hidden)
    grads = []
    for mode in ('auto', 'f32'):
        monkeypatch.setenv('NBASR_DENSE_MODE', mode)
        ps = [p.clone().to(DEV).requires_grad_(True) for p in params]
        xg = x.to(DEV).requires_grad_(True)
        (nb_autograd.lstm(xg, *ps) * r.to(DEV)).sum().backward()
        grads.append([xg.grad] + [p.grad for p in ps])
    for got, want in zip(*grads):
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert not all(torch.equal(g, w) for g, w in zip(*grads))


@pytest.mark.parametrize('c,hidden,b,t', [(24, 8, 2, 5), (40, 20, 3, 17), (1200, 500, 2, 9), (16, 4, 5, 1)])
def test_lstm_bptt_vs_torch_autograd(c, hidden, b, t):
    """BPTT of the single-layer LSTM against ATen's autograd of nn.LSTM (the reference's own module, model.py:100) on the CPU:
    gradients with respect to the input and all four parameters, gate order i, f, g, o."""
    torch.manual_seed(c + hidden + t)
    ref = torch.nn.LSTM(c, hidden, batch_first=True)
    x = torch.randn(b, c, t) * 1.5
    r = torch.randn(b, t, hidden)
    xr = x.clone().requires_grad_(True)
    out, _ = ref(xr.permute(0, 2, 1))
    (out * r).sum().backward()
    params = [p.detach().clone().to(DEV).requires_grad_(True) for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    xg = x.to(DEV).requires_grad_(True)
    h = nb_autograd.lstm(xg, *params)
    assert torch.allclose(h.detach().cpu(), out.detach(), rtol=1e-4, atol=1e-5)
    (h * r.to(DEV)).sum().backward()
    close(xg.grad, xr.grad.numpy(), 'dx', rtol=2e-4)
    for got, want, name in zip(params, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0), ('dw_ih', 'dw_hh', 'db_ih', 'db_hh')):
        close(got.grad, want.grad.numpy(), name, rtol=2e-4)
