"""Host logic of the executor that needs no GPU: launch-tape argument patching, the node-kernel variant table, tape keys."""
import json
import pathlib

import pytest
import torch
import torch.nn as nn

import cases
import nb_asr_amd as nb
from nb_asr_amd import executor, hip


class _Ptr:
    """Stands in for a tensor whose address a tape patches."""

    def __init__(self, p):
        self.p = p

    def data_ptr(self):
        return self.p


def test_launch_tape_patches_the_input_and_produced_tensors():
    calls = []

    def fn_a(*args):
        calls.append(('a', args))
        return 0

    def fn_b(*args):
        calls.append(('b', args))
        return 0
    fn_a.__name__, fn_b.__name__ = 'fn_a', 'fn_b'
    produced = iter([_Ptr(0x9000), _Ptr(0xA000)])
    x0, logits0 = 0x1000, 0x7000
    entries = [
        [fn_a, [x0, 0x2000, 64, 3, None, 0.5]],                      # reads the input
        [None, lambda: next(produced), logits0, []],                 # host step that allocates the logits
        [fn_b, [0x2000, logits0, 49, x0]],                           # writes them (and reads x again)
        [fn_a, [0x3000, logits0, 7, 1, None, 1.0]],
    ]
    tape = executor.LaunchTape(entries, x0, pipelined=False, pipe=False)
    assert tape.x_slots == [(0, 0), (2, 3)] and entries[1][3] == [(2, 1), (3, 1)] and tape.launches == 3

    class Plan:
        _tape_logits = 'logits'
        tail_done = [None, None]
        _turn = 0

        def wait_tails(self):
            calls.append(('wait', ()))
    plan = Plan()
    assert tape.replay(plan, _Ptr(0x5000)) == 'logits' and plan._tape_logits is None
    assert calls == [('wait', ()), ('a', (0x5000, 0x2000, 64, 3, None, 0.5)), ('b', (0x2000, 0x9000, 49, 0x5000)),
                     ('a', (0x3000, 0x9000, 7, 1, None, 1.0))]
    calls.clear()
    plan._tape_logits = 'again'
    tape.replay(plan, _Ptr(0x6000))
    assert calls[1] == ('a', (0x6000, 0x2000, 64, 3, None, 0.5)) and calls[2] == ('b', (0x2000, 0xA000, 49, 0x6000))
    # small integers that happen to equal nothing are never patched: 64, 3, 49, 7 above stayed what they were


def test_launch_tape_reports_a_failing_call(monkeypatch):
    def bad(*args):
        return -1
    bad.__name__ = 'nbasr_something'
    monkeypatch.setattr(hip, '_check', lambda rc, what: (_ for _ in ()).throw(hip.HipError(f'{what} failed with code {rc}')))
    tape = executor.LaunchTape([[bad, [1, 2]]], 0x1000, False, False)

    class Plan:
        _tape_logits = None
        tail_done = [None, None]
        _turn = 0

        def wait_tails(self):
            pass
    with pytest.raises(hip.HipError, match='nbasr_something failed with code -1'):
        tape.replay(Plan(), _Ptr(0x1000))


def test_variant_table_is_complete_and_well_formed():
    table = json.loads(pathlib.Path(executor.__file__).with_name('gc_variant_table.json').read_text())
    # (taps, dilation) x channels per group x {4 flavours + 2 statistics flavours} x {small, large}
    assert table == executor._GC_TABLE and len(table) == 4 * 4 * (4 + 2) * 2
    no_split = (0, hip.GC_PIPE, hip.GC_RING)
    for key, v in table.items():
        parts = key.split(',')
        k, d, cg = map(int, parts[:3])
        assert (k, d) in ((5, 1), (5, 2), (7, 1), (7, 2)) and cg in (6, 8, 10, 12)
        assert parts[3] in ('lnx', 'lnx+skip', 'skip', 'plain') and parts[4] in ('small', 'large')
        if len(parts) == 6:
            assert parts[5] == 'stats' and parts[3] in ('plain', 'skip') and v in no_split      # statistics launches: no output split
        else:
            assert v in no_split + (hip.GC_OSPLIT, hip.GC_PIPE | hip.GC_OSPLIT)
    # every key has been measured in profiles/: the table is a function of the committed logs
    logs = list((pathlib.Path(__file__).resolve().parent.parent / 'profiles' / 'r03_gc_variants').glob('*.jsonl'))
    rows = [json.loads(line) for f in logs for line in f.read_text().splitlines() if line.startswith('{')]
    assert len([r for r in rows if r['batch'] in (8, 64)]) == len(table)


def test_variant_choice_per_launch(monkeypatch):
    plan = executor.ForwardPlan('cpu')
    model = nb.get_model(cases.ARCH_D, use_rnn=True, dropout_rate=0.0)
    cell = next(m for m in model.model if type(m).__name__ == 'SearchCell' and m.filters == 1200)
    node = cell.nodes[0]                                           # conv7d2 or similar with 12 channels per group
    op = node.op
    cg = 1200 // op.groups

    def view(b, frames):
        return torch.empty(1).as_strided((b, 1200, hip.round_up4(frames)), (0, 0, 0))        # only its shape is looked at
    has_skip = any(type(br).__name__ == 'Identity' for br in node.branch_ops)
    flavour = 'skip' if has_skip else 'plain'
    for b, frames, size in ((64, 250, 'large'), (8, 250, 'small'), (2, 40, 'small')):
        want = executor._GC_TABLE[f'{op.kernel_size},{op.dilation},{cg},{flavour},{size}']
        assert plan._gc_variant(view(b, frames), node, None, None, 2) == want
        assert plan._gc_variant(view(b, frames), node, None, ('stats',), 2) == executor._GC_TABLE[f'{op.kernel_size},{op.dilation},{cg},{flavour},{size},stats']
    ln0 = ('stats', 'gamma', 'beta')
    assert plan._gc_variant(view(64, 250), node, ln0, None, 1) == executor._GC_TABLE[f'{op.kernel_size},{op.dilation},{cg},{"lnx+skip" if has_skip else "lnx"},large']
    assert plan._gc_variant(view(64, 250), None) == 0                                   # no node: the default kernel
    monkeypatch.setenv('NBASR_GC_F32_VARIANT', str(hip.GC_PIPE | hip.GC_OSPLIT))
    assert plan._gc_variant(view(64, 250), node, None, None, 2) == hip.GC_PIPE | hip.GC_OSPLIT
    assert plan._gc_variant(view(64, 250), node, None, ('stats',), 2) == hip.GC_PIPE       # the split is never forced onto a statistics launch
    monkeypatch.delenv('NBASR_GC_F32_VARIANT')
    monkeypatch.setenv('NBASR_GC_F32_VARIANT', '0')                                     # the default kernel everywhere
    assert executor.ForwardPlan('cpu')._gc_variant(view(64, 250), node, None, None, 2) == 0


def test_structure_epoch_counts_registrations():
    before = executor._structure_epoch[0]
    m = nn.Linear(3, 3)
    assert executor._structure_epoch[0] > before                    # weight and bias were registered
    mid = executor._structure_epoch[0]
    seq = nn.Sequential(m)
    seq[0] = nn.Linear(3, 3)
    assert executor._structure_epoch[0] > mid


def test_bench_reads_only_attributes_a_plan_has():
    """bench.py reports bookkeeping of the last plan in its JSON line; a removed plan attribute must not survive there (round 3 pruned
    several switches -- the bench line is produced on the GPU box only, so this is checked here, without a GPU)."""
    import re
    root = pathlib.Path(__file__).resolve().parent.parent
    plan = executor.ForwardPlan('cpu')
    for name in sorted(set(re.findall(r'\bplan\.([A-Za-z_]+)', (root / 'bench.py').read_text()))):
        assert hasattr(plan, name), f'bench.py reads plan.{name}, which ForwardPlan does not have'


def test_removed_environment_switches_are_announced_once(monkeypatch):
    """a script that still sets a switch of rounds 1-3 (NBASR_TRAIN_GEMM became NBASR_DENSE_MODE, ...) is told so, once."""
    import warnings
    from nb_asr_amd import executor
    monkeypatch.setenv('NBASR_TRAIN_GEMM', 'f32')
    monkeypatch.setattr(executor, '_warned_removed', [False])
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter('always')
        executor._warn_removed_switches()
        executor._warn_removed_switches()
    assert len(seen) == 1 and 'NBASR_TRAIN_GEMM' in str(seen[0].message) and 'NBASR_DENSE_MODE' in str(seen[0].message)
