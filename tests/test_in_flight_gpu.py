"""Several chains of forwards in flight on several streams of ONE device, one host thread per stream (round 6: `bench.py --in-flight 2`,
the form small batches are served in -- 8 utterances occupy 200 of 256 compute units with one wave per SIMD, and a second chain's
kernels fill the rest).  What must hold: every forward's logits are those of a lone forward, bit for bit.

Two things broke this before they were found with tools/ubench/in_flight_check.py:
  * executor.PlanPool handed a plan back as soon as its forward was ENQUEUED and gave it to whichever thread asked next -- a thread on
    another stream then wrote the workspaces the first stream was still reading;
  * the cached per-frame recurrence graph zeroed its workspace with hipMemsetAsync (a memset NODE of the graph), which ran out of order
    with three chains in flight and the first of them on the default stream: NaN logits in every second forward.
"""
import threading

import pytest
import torch

import nb_asr_amd as nb
from nb_asr_amd.weights import keyed_fill_, keyed_input

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _model():
    model = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
    keyed_fill_(model, seed=1235, mode='lively')
    return model.to(DEV).eval()


def _run_chains(model, xs, want, steps, overlap_main, plain=False):
    ways = len(xs)
    streams = [torch.cuda.Stream(device=DEV) for _ in range(ways)]
    bad, errors = [[] for _ in range(ways)], []

    def worker(i):
        try:
            with torch.no_grad(), torch.cuda.stream(streams[i]):
                if plain:
                    outs = [model(xs[i]) for _ in range(steps)]
                else:
                    outs = [h.result() for h in [model.forward_async(xs[i]) for _ in range(steps)]]
                streams[i].synchronize()
                bad[i] = [k for k, o in enumerate(outs) if not torch.equal(o, want[i])]
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    for _round in range(3):             # the recurrence chains become cached graphs after their third use: round 0 records, 1-2 replay
        main_outs = None
        if overlap_main:
            with torch.no_grad():
                main_outs = [h.result() for h in [model.forward_async(xs[0]) for _ in range(steps)]]      # default stream, not waited for
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(ways)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert all(not b for b in bad), (_round, bad)
        if main_outs is not None:
            torch.cuda.synchronize()
            assert all(torch.equal(o, want[0]) for o in main_outs), _round


@pytest.mark.parametrize('batch,frames,ways,overlap_main', [(8, 1000, 2, True), (8, 1000, 2, False), (16, 400, 3, True), (3, 333, 2, True)])
def test_chains_in_flight_on_several_streams_return_the_lone_forwards_logits(batch, frames, ways, overlap_main):
    model = _model()
    xs = [keyed_input(batch, frames, seed=i).to(DEV) for i in range(ways)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    _run_chains(model, xs, want, steps=24, overlap_main=overlap_main)
    # one plan per stream that ran forwards (+ the default stream's): a plan stays with the stream it was released on
    assert len(model._plans) == ways + 1


def test_plain_forwards_from_two_threads_on_two_streams():
    model = _model()
    xs = [keyed_input(8, 500, seed=i).to(DEV) for i in range(2)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    _run_chains(model, xs, want, steps=10, overlap_main=False, plain=True)


def test_a_plan_changes_streams_only_behind_its_release_event():
    """More streams than the pool keeps plans for: the oldest idle plan is handed over, behind the event its release recorded."""
    from nb_asr_amd.executor import PlanPool
    model = _model()
    x = keyed_input(4, 200, seed=1).to(DEV)
    with torch.no_grad():
        want = model(x).clone()
        outs = []
        for _ in range(PlanPool.MAX_IDLE_PER_DEVICE + 3):
            s = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(s):
                outs.append((s, model.forward_async(x).result()))
        for s, o in outs:
            s.synchronize()
            assert torch.equal(o, want)
    assert len(model._plans) <= PlanPool.MAX_IDLE_PER_DEVICE


@pytest.mark.parametrize('in_flight', [1, 2, 3])
def test_forward_many_returns_every_batchs_own_logits_in_order(in_flight):
    """`ASRModel.forward_many`: batches of different sizes and lengths, W chains in flight -- each result is that batch's lone forward."""
    model = _model()
    shapes = [(8, 1000), (3, 333), (8, 1000), (5, 64), (8, 1000), (1, 17), (8, 1000), (8, 999), (2, 500), (8, 1000), (8, 1000), (4, 250)]
    xs = [keyed_input(b, t, seed=10 + i).to(DEV) for i, (b, t) in enumerate(shapes)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    for _ in range(3):
        got = model.forward_many(xs, in_flight=in_flight)
        assert len(got) == len(xs)
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    assert model.forward_many([], in_flight=in_flight) == []


def test_forward_many_raises_in_the_callers_thread():
    model = _model()
    with pytest.raises(ValueError, match='expected a'):
        model.forward_many([torch.zeros(2, 79, 50, device=DEV)] * 4, in_flight=2)


@pytest.mark.parametrize('in_flight,tail_group', [(1, 2), (1, 'auto'), (2, 'auto'), (2, 3), (2, 8), (2, 1)])
def test_tail_groups_share_one_recurrence_and_change_no_bit(in_flight, tail_group):
    """`forward_many(tail_group=...)`: consecutive batches of one shape in a chain run their LSTM + head as ONE recurrence over all
    their utterances (the projection writes each forward's gates into its rows of the group's gate tensor); an utterance's result does
    not depend on the batch it is computed in, so every forward's logits are the lone forward's.  Mixed shapes break the runs."""
    model = _model()
    shapes = [(8, 1000)] * 9 + [(3, 333)] * 4 + [(8, 1000)] * 5 + [(5, 64)] + [(8, 1000)] * 6 + [(4, 250)] * 7
    xs = [keyed_input(b, t, seed=10 + i).to(DEV) for i, (b, t) in enumerate(shapes)]
    with torch.no_grad():
        want = [model(x).clone() for x in xs]
    torch.cuda.synchronize()
    for _ in range(3):                  # first pass records the launch tapes of every (member, slot), later passes replay them
        got = model.forward_many(xs, in_flight=in_flight, tail_group=tail_group)
        assert len(got) == len(xs)
        for i, (g, w) in enumerate(zip(got, want)):
            assert g.shape == w.shape and torch.equal(g, w), i


def test_a_tail_group_members_result_needs_the_whole_group():
    model = _model()
    x = keyed_input(4, 200, seed=1).to(DEV)
    with torch.no_grad():
        first = model.forward(x, _pipelined=True, _group=(0, 2))
        with pytest.raises(nb.hip.HipError, match='tail group'):
            first.result()
        second = model.forward(x, _pipelined=True, _group=(1, 2))
        want = model(x)
        assert torch.equal(first.result(), want) and torch.equal(second.result(), want)


def test_chosen_streams_run_side_by_side_and_are_remembered():
    """streams.py: the chain streams of forward_many are pairwise on different hardware queues, a main stream's tail stream overlaps
    with it and is the same one at every request (a plan that is dropped and built again runs its tail where it ran before)."""
    from nb_asr_amd import streams
    chains = streams.chain_streams(DEV, 2)
    assert len({s.cuda_stream for s in chains}) == 2
    assert streams.overlaps(DEV, chains[0], chains[1])
    for s in chains:
        tail = streams.tail_stream_for(DEV, s)
        assert tail.cuda_stream != s.cuda_stream and streams.overlaps(DEV, tail, s)
        assert streams.tail_stream_for(DEV, s) is tail
    assert not streams.overlaps(DEV, chains[0], chains[0])
    default_tail = streams.tail_stream_for(DEV, torch.cuda.default_stream(DEV))
    assert streams.overlaps(DEV, default_tail, torch.cuda.default_stream(DEV))
    model = _model()
    x = keyed_input(4, 200, seed=2).to(DEV)
    with torch.no_grad():
        want = model(x).clone()
        assert torch.equal(model.forward_async(x).result(), want)
        side = model._plans.values()[-1].side_stream
        model._plans.clear()
        assert torch.equal(model.forward_async(x).result(), want)
        assert model._plans.values()[-1].side_stream is side
