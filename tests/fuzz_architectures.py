#!/usr/bin/env python3
"""Extended random sweep (diagnostic; the committed tests run 8 of these): random architectures x random (B, T) x both heads
x both initialisations, HIP forward vs the CPU oracle with the noise-aware tolerance of tests/test_model_gpu.py, plus the
pipelined path.  Lives under tests/ because it uses the oracle (test infrastructure).  usage: python tests/fuzz_architectures.py [N=40] [first_seed=100]"""
import pathlib, random, sys
import torch
root = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / 'tests'))
import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input
from oracle import asr_oracle as oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst, bad = 0.0, 0
for seed in range(first, first + n):
    rng = random.Random(seed)
    arch = nb.get_random_architectures(1, seed=9000 + seed)[0]
    use_rnn = rng.random() < 0.6
    b, t = rng.choice([1, 2, 3, 5]), rng.choice([1, 3, 4, 5, 31, 64, 97, 130, 201, 255, 256, 257, 300, 513])
    mode = rng.choice(['lively', 'lively', 'xavier'])
    m = keyed_fill_(nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0), seed=seed, mode=mode).to('cuda:0').eval()
    x = keyed_input(b, t, seed=seed)
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    want = oracle.asr_forward(params, arch, x, use_rnn=use_rnn)
    truth = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to('cuda:0')).cpu()
        got2 = m.forward_async(x.to('cuda:0')).result().cpu() if use_rnn else got
    noise = cases.worst_ratio(want, truth, 1e-4, 1e-5)
    ratio = cases.worst_ratio(got, want, 1e-4, 1e-5)
    scale = float(want.abs().max())
    rel = float((got - want).abs().max()) / max(scale, 1e-300)
    try:                                   # the two-leg rule of tests/cases.py: un-relaxed bound, or "no further from fp64 than the oracle"
        cases.assert_parity(got, want, truth, f'seed {seed}')
        within = True
    except AssertionError as exc:
        within = False
        print('   ', exc)
    ok = within and torch.equal(got, got2) and bool(torch.isfinite(got).all())
    worst = max(worst, ratio if noise < cases.QUIET else cases.worst_ratio(got, truth, 1e-4, 1e-5) / (1.5 * noise))
    bad += not ok
    print(f'{"ok " if ok else "BAD"} seed {seed} arch {arch} rnn={int(use_rnn)} {mode:6s} b={b} t={t:3d}: err/tol {ratio:6.3f} noise {noise:5.3f} rel {rel:.1e} scale {scale:.1e}', flush=True)
print(f'{n} cases, {bad} failures, worst margin use {worst:.2f}')
sys.exit(1 if bad else 0)
