#!/usr/bin/env python3
"""Random sweep: random architectures x random (B, T) x both heads x both initialisations, HIP forward vs the CPU oracle under the
two-leg parity rule of tests/cases.py, plus the pipelined path (bit-equal to the plain one).  Lives under tests/ because it uses
the oracle (test infrastructure).  `tests/test_fuzz_gpu.py` collects a fixed seed list of it under `-m gpu`; as a script it runs
any range:   python tests/fuzz_architectures.py [N=40] [first_seed=100] [--strict]
(--strict: every GEMM on the exact-fp32 MFMA, NBASR_DENSE_MODE=f32 NBASR_LINEAR_MODE=f32, for a default-vs-strict comparison)."""
import os
import pathlib
import random
import sys

import torch

root = pathlib.Path(__file__).resolve().parent.parent
for p in (str(root), str(root / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import cases                                                  # noqa: E402
import nb_asr_amd as nb                                       # noqa: E402
from nb_asr_amd.weights import keyed_fill_, keyed_input       # noqa: E402
from oracle import asr_oracle as oracle                       # noqa: E402

FRAMES = [1, 3, 4, 5, 31, 64, 97, 130, 201, 255, 256, 257, 300, 513]


def case_of(seed):
    """(arch, use_rnn, batch, frames, init mode) of a seed -- the derivation every recorded sweep used (profiles/r02_fuzz_*)."""
    rng = random.Random(seed)
    arch = nb.get_random_architectures(1, seed=9000 + seed)[0]
    use_rnn = rng.random() < 0.6
    b, t = rng.choice([1, 2, 3, 5]), rng.choice(FRAMES)
    mode = rng.choice(['lively', 'lively', 'xavier'])
    return arch, use_rnn, b, t, mode


def run_case(seed, device='cuda:0'):
    """One case on the default path.  Returns a dict: ok, why (the violated leg, if any), the numbers of the record line."""
    arch, use_rnn, b, t, mode = case_of(seed)
    m = keyed_fill_(nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0), seed=seed, mode=mode).to(device).eval()
    x = keyed_input(b, t, seed=seed)
    params = {k: v.cpu() for k, v in m.state_dict().items()}
    want = oracle.asr_forward(params, arch, x, use_rnn=use_rnn)
    truth = oracle.asr_forward(params, arch, x, use_rnn=use_rnn, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(device)).cpu()
        got2 = m.forward_async(x.to(device)).result().cpu() if use_rnn else got
    noise = cases.worst_ratio(want, truth, 1e-4, 1e-5)
    ratio = cases.worst_ratio(got, want, 1e-4, 1e-5)
    scale = float(want.abs().max())
    rel = float((got - want).abs().max()) / max(scale, 1e-300)
    why = ''
    try:                                   # the two-leg rule of tests/cases.py: un-relaxed bound, or "no further from fp64 than the oracle"
        cases.assert_parity(got, want, truth, f'seed {seed}')
    except AssertionError as exc:
        why = str(exc)
    if not why and not torch.equal(got, got2):
        why = f'seed {seed}: the pipelined forward differs from the plain one'
    if not why and not bool(torch.isfinite(got).all()):
        why = f'seed {seed}: non-finite logits'
    rms_ratio = cases._rms(got.double() - truth) / max(cases._rms(want.double() - truth), 1e-300)
    margin = ratio if noise < cases.QUIET else cases.worst_ratio(got, truth, 1e-4, 1e-5) / (1.5 * noise)
    line = (f'{"ok " if not why else "BAD"} seed {seed} arch {arch} rnn={int(use_rnn)} {mode:6s} b={b} t={t:3d}: err/tol {ratio:6.3f} '
            f'noise {noise:5.3f} rms/ref {rms_ratio:5.2f} rel {rel:.1e} scale {scale:.1e}')
    return dict(ok=not why, why=why, line=line, margin=margin, rms_ratio=rms_ratio, noise=noise, ratio=ratio)


def main(argv):
    args = [a for a in argv if not a.startswith('--')]
    if '--strict' in argv:
        os.environ['NBASR_DENSE_MODE'], os.environ['NBASR_LINEAR_MODE'] = 'f32', 'f32'
    n = int(args[0]) if args else 40
    first = int(args[1]) if len(args) > 1 else 100
    worst, bad, ratios = 0.0, 0, []
    for seed in range(first, first + n):
        r = run_case(seed)
        if r['why']:
            print('   ', r['why'])
        print(r['line'], flush=True)
        worst, bad = max(worst, r['margin']), bad + (not r['ok'])
        ratios.append(r['rms_ratio'])
    ratios.sort()
    print(f'{n} cases, {bad} failures, worst margin use {worst:.2f}; rms error vs fp64 relative to the reference\'s: '
          f'median {ratios[len(ratios) // 2]:.2f}, max {ratios[-1]:.2f}')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
