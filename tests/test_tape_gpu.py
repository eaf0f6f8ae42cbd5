"""Launch tapes (executor.LaunchTape): a replayed forward must be BIT-IDENTICAL to the python launch sequence, whatever
changes between calls -- the input tensor, the logits' address (allocator churn), the pipelined slot, the stream, the
parameters (in-place update, swapped storage), the workspace addresses (growth)."""
import pytest
import torch

import cases
import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build(arch, use_rnn, dtype=torch.float32, seed=31):
    m = nb.get_model(arch, use_rnn=use_rnn, dropout_rate=0.0)
    keyed_fill_(m, seed=seed, mode='lively')
    return m.to(DEV).to(dtype).eval()


def plans(m):
    return list(m._plans.values())


def untaped(m, xs, monkeypatch, call=lambda m, x: m(x)):
    monkeypatch.setenv('NBASR_TAPE', '0')
    m._plans.clear()
    with torch.no_grad():
        out = [call(m, x).clone() for x in xs]
    assert all(p.tape_replays == 0 and not p._tapes for p in plans(m))
    monkeypatch.delenv('NBASR_TAPE')
    m._plans.clear()
    return out


def churn(i):
    """Move the caching allocator's free lists so that the next logits land at another address."""
    keep = [torch.empty(1000 + 517 * i + 64 * j, device=DEV) for j in range(3)]
    return keep[i % 3]


@pytest.mark.parametrize('arch,use_rnn,b,t,dtype', [
    (cases.ARCH_D, True, 3, 131, torch.float32), (cases.ARCH_M, False, 2, 258, torch.float32),
    ([[2, 1], [3, 0, 1], [4, 1, 0, 1]], True, 2, 64, torch.float32), (cases.ARCH_D, True, 2, 96, torch.bfloat16),
    (cases.ARCH_M, False, 2, 75, torch.bfloat16)])
def test_replay_is_bit_identical(monkeypatch, arch, use_rnn, b, t, dtype):
    m = build(arch, use_rnn, dtype)
    xs = [keyed_input(b, t, seed=s).to(DEV).to(dtype) for s in range(6)]
    want = untaped(m, xs, monkeypatch)
    held = []
    with torch.no_grad():
        for i, x in enumerate(xs):
            held.append(churn(i))
            got = m(x)
            assert torch.equal(got, want[i]), f'call {i}'
    (plan,) = plans(m)
    assert plan.tape_replays == len(xs) - 2 and len(plan._tapes) == 1      # call 0 python, call 1 recorded, the rest replayed
    tape = next(iter(plan._tapes.values()))
    assert tape.x_slots and tape.launches > 20


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pipelined_replay_is_bit_identical(monkeypatch, dtype):
    m = build(cases.ARCH_D, True, dtype)
    xs = [keyed_input(3, 140, seed=s).to(DEV).to(dtype) for s in range(9)]
    want = untaped(m, xs, monkeypatch)
    with torch.no_grad():
        handles = []
        for i, x in enumerate(xs):
            handles.append(m.forward_async(x))
            churn(i)
            if i == 4:                                  # a plain forward in between shares the tail's buffers
                assert torch.equal(m(xs[0]), want[0])
        for i, h in enumerate(handles):
            assert torch.equal(h.result(), want[i]), f'call {i}'
    (plan,) = plans(m)
    # one tape per pipelined slot (the plain forward in between may grow a workspace of its own, which drops the tapes once)
    assert plan.tape_replays >= 3 and len(plan._tapes) >= 2


def test_parameter_changes_invalidate_the_tape(monkeypatch):
    m = build(cases.ARCH_D, True)
    x = keyed_input(2, 120, seed=3).to(DEV)
    with torch.no_grad():
        for _ in range(3):
            y0 = m(x)
        (plan,) = plans(m)
        assert plan.tape_replays == 1
        m.model[0].conv.weight.mul_(1.5)                # in-place update: version counter
        y1 = m(x)
        assert not torch.equal(y1, y0)
        head = m.model[-1]
        head.weight.data = head.weight.data * 0.5      # swapped storage: address
        ys = [m(x) for _ in range(3)]
        assert plan.tape_replays >= 2                   # re-recorded for the new parameters, then replayed again
    want = untaped(m, [x], monkeypatch)[0]
    assert all(torch.equal(y, want) for y in ys) and not torch.equal(y1, want)


def test_growth_and_other_shapes(monkeypatch):
    m = build(cases.ARCH_D, True)
    small = [keyed_input(2, 64, seed=s).to(DEV) for s in range(4)]
    big = [keyed_input(4, 200, seed=s).to(DEV) for s in range(3)]
    ragged = [keyed_input(2, 63, seed=s).to(DEV) for s in range(3)]
    want = untaped(m, small + big + ragged, monkeypatch)
    with torch.no_grad():
        got = [m(x) for x in small[:3]] + [m(x) for x in big] + [m(small[3])] + [m(x) for x in ragged]
    order = list(range(3)) + [4, 5, 6] + [3] + [7, 8, 9]
    for g, k in zip(got, order):
        assert torch.equal(g, want[k]), k


def test_other_stream_and_misaligned_input(monkeypatch):
    m = build(cases.ARCH_D, True)
    xs = [keyed_input(2, 100, seed=s).to(DEV) for s in range(4)]
    want = untaped(m, xs, monkeypatch)
    side = torch.cuda.Stream()
    with torch.no_grad():
        for i in range(3):
            assert torch.equal(m(xs[i]), want[i])
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(4):
                assert torch.equal(m(xs[i]), want[i])
        torch.cuda.current_stream().wait_stream(side)
        # an input that does not start on a 16-byte boundary takes the re-pitching path: its own key, same numbers
        flat = torch.empty(xs[3].numel() + 1, device=DEV)
        odd = flat[1:].view_as(xs[3])
        odd.copy_(xs[3])
        for _ in range(3):
            assert torch.equal(m(odd), want[3])


def test_swapped_modules_are_seen(monkeypatch):
    """A layer replaced inside a model that has already run (fine-tuning a new head, a re-initialised node op) must reach the
    next forward: module / parameter registration anywhere invalidates the recorded tapes."""
    import torch.nn as nn
    m = build(cases.ARCH_D, True)
    x = keyed_input(2, 80, seed=4).to(DEV)
    with torch.no_grad():
        for _ in range(3):
            y0 = m(x)
        (plan,) = plans(m)
        assert plan.tape_replays == 1
        torch.manual_seed(3)
        m.model[-1] = nn.Linear(500, 49).to(DEV)                        # a new head
        y1 = m(x)
        op = m.model[2].nodes[0].op                                     # a node op deep inside a cell
        op.conv = nn.Conv1d(op.conv.in_channels, op.conv.out_channels, op.kernel_size, dilation=op.dilation, groups=op.groups).to(DEV)
        ys = [m(x) for _ in range(3)]
    assert not torch.equal(y1, y0) and not torch.equal(ys[0], y1)
    want = untaped(m, [x], monkeypatch)[0]
    assert all(torch.equal(y, want) for y in ys)


def test_plain_attribute_mutations_are_seen(monkeypatch):
    """`node.branch_ops` is a python list and `cell.use_norm` a plain attribute -- no registration hook fires when they
    change, so the tape key carries a structural fingerprint.  After a tape exists, flipping a skip flag or dropping a cell's
    LayerNorm must reach the next forward (the pre-tape executor re-read these on every call)."""
    from nb_asr_amd.ops import Identity, Zero
    m = build(cases.ARCH_D, True)
    x = keyed_input(2, 80, seed=5).to(DEV)
    with torch.no_grad():
        for _ in range(3):
            y0 = m(x)
        (plan,) = plans(m)
        assert plan.tape_replays == 1
        node = m.model[3].nodes[1]
        assert isinstance(node.branch_ops[0], Identity)
        node.branch_ops[0] = Zero()                                       # a skip connection removed
        ys = [m(x) for _ in range(3)]
        m.model[4].use_norm = False                                       # a cell without its LayerNorm
        zs = [m(x) for _ in range(3)]
    assert not torch.equal(ys[0], y0) and not torch.equal(zs[0], ys[0])
    assert all(torch.equal(y, ys[0]) for y in ys) and all(torch.equal(z, zs[0]) for z in zs)
    want = untaped(m, [x], monkeypatch)[0]
    assert torch.equal(zs[-1], want)
    m.model[4].use_norm = True
    node.branch_ops[0] = Identity()
    with torch.no_grad():
        assert torch.equal(m(x), y0)
