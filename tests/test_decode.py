"""Validation decode: CTC prefix beam search, phoneme folding, token error rate.

CPU part: the oracle against the reference's own folding output (tests/golden/decode_fixtures.npz), against exhaustive CTC
enumeration and against known answers; the package's phoneme table against the reference's data.
GPU part: the HIP kernels, through the C ABI, against the oracle."""
import pathlib

import numpy as np
import pytest
import torch

from oracle import decode_oracle as oracle

GOLDEN = np.load(pathlib.Path(__file__).parent / 'golden' / 'decode_fixtures.npz')


def _rows():
    return list(zip(GOLDEN['rows61'].tolist(), GOLDEN['rows48'].tolist(), GOLDEN['rows39'].tolist()))


def _log_probs(shape, seed, sharp=2.0):
    gen = torch.Generator().manual_seed(seed)
    return torch.log_softmax(torch.randn(*shape, generator=gen) * sharp, dim=-1)


# ---- oracle pins ----------------------------------------------------------------------------------------------------------

def test_oracle_folding_matches_reference_output():
    rows = _rows()
    lists = oracle.class_lists(rows)
    assert lists[1] == GOLDEN['vocab48'].tolist() and lists[2] == GOLDEN['vocab39'].tolist()
    mapping = oracle.index_mapping(rows, 1, 2)
    assert [mapping[i] for i in range(49)] == GOLDEN['map48to39'].tolist()
    assert np.array_equal(oracle.fold_encoded(GOLDEN['seq_in'], mapping), GOLDEN['seq_out'])
    table = oracle.fold_table(mapping)
    assert np.array_equal(table[GOLDEN['seq_in']], GOLDEN['seq_out'])
    # the reference's relabelling order is visible in its own output: 'cl' does not end as 'sil'
    v48, v39 = ['_'] + lists[1], ['_'] + lists[2]
    assert v39[table[v48.index('cl')]] == 'ng' and v39[mapping[v48.index('cl')]] == 'sil'


def test_package_phoneme_table_matches_reference_data():
    from nb_asr_amd import phonemes
    assert [list(r) for r in phonemes.FOLDING] == [list(r) for r in _rows()]
    assert phonemes.vocab(48) == GOLDEN['vocab48'].tolist() and phonemes.vocab(39) == GOLDEN['vocab39'].tolist()
    assert phonemes.vocab(48, inc_blank=True) == GOLDEN['vocab48_blank'].tolist()
    assert phonemes.fold_table(48, 39, sequential=False).tolist() == GOLDEN['map48to39'].tolist()
    table = phonemes.fold_table(48, 39)
    assert np.array_equal(table.numpy()[GOLDEN['seq_in']], GOLDEN['seq_out'])
    assert len(phonemes.vocab(61)) == 61 and phonemes.fold_table(61, 39, sequential=False).max() == 39
    with pytest.raises(ValueError):
        phonemes.index_mapping(39, 48)


@pytest.mark.parametrize('seed', range(12))
def test_oracle_beam_search_is_exhaustive_ctc_when_wide(seed):
    rng = np.random.default_rng(seed)
    frames, classes = int(rng.integers(1, 7)), int(rng.integers(2, 5))
    lp = _log_probs((frames, classes), seed).numpy()
    exact = oracle.ctc_labelling_log_probs(lp)
    beams = oracle.ctc_beam_search(lp, beam_width=4096)
    got = {tuple(tok): -score for tok, score in beams if score < 1e38}      # prefixes no alignment reaches keep score -FLT_MAX
    assert set(got) == set(exact)
    for lab, v in exact.items():
        assert abs(got[lab] - v) < 1e-4
    assert tuple(beams[0][0]) == max(exact, key=exact.get)
    assert [sc for _, sc in beams] == sorted(sc for _, sc in beams)


def test_oracle_beam_search_hand_cases():
    def onehot(seq, classes=5):
        x = np.full((len(seq), classes), -20.0, dtype=np.float32)
        for t, c in enumerate(seq):
            x[t, c] = 0.0
        return x
    assert oracle.ctc_beam_search(onehot([0, 1, 1, 0, 1, 2, 2, 0, 0, 3]), 4)[0][0] == [1, 1, 2, 3]
    assert oracle.ctc_beam_search(onehot([0, 0, 0]), 4)[0][0] == []
    assert oracle.ctc_beam_search(onehot([2, 2, 2, 2]), 1)[0][0] == [2]
    assert oracle.ctc_beam_search(np.zeros((0, 5), dtype=np.float32), 3) == [([], -0.0)]
    # two alignments of 'a' together beat the single most probable path (blank, blank): where greedy and beam search differ
    lp = np.log(np.array([[0.6, 0.4], [0.6, 0.4]], dtype=np.float32))
    assert oracle.ctc_beam_search(lp, 2)[0][0] == [1]
    # pruning to the top classes: with cutoff_top_n = 1 only the frame's best class is ever extended
    lp = _log_probs((6, 7), 3).numpy()
    best = [int(c) for c in lp.argmax(axis=1)]
    collapsed = [c for i, c in enumerate(best) if c != 0 and (i == 0 or c != best[i - 1])]
    assert oracle.ctc_beam_search(lp, 5, cutoff_top_n=1)[0][0] == collapsed


def test_oracle_edit_distance_known_answers():
    def dist(a, b):
        return oracle.edit_distance(list(a), list(b))
    assert dist('kitten', 'sitting') == 3 and dist('', 'abc') == 3 and dist('abc', '') == 3 and dist('', '') == 0
    assert dist('flaw', 'lawn') == 2 and dist('intention', 'execution') == 5 and dist('abc', 'abc') == 0
    counts = oracle.error_counts([[1, 0, 2, 2, 9], [0, 0, 0]], [4, 3], [[1, 2, 0, 3], [4, 0, 0, 0]], [4, 1])
    assert counts == [(1, 3), (1, 1)]               # [1,2,2] vs [1,2,3]: one substitution; [] vs [4]: one deletion
    table = [0, 1, 2, 2, 0, 5, 6, 7, 8, 3]
    assert oracle.error_counts([[1, 0, 2, 2, 9]], [5], [[1, 2, 0, 3]], [4], table=table) == [(1, 3)]    # [1,2,2,3] vs [1,2,2]


# ---- HIP kernels against the oracle ---------------------------------------------------------------------------------------

def _check_beams(log_probs, lengths, width, top_n, got):
    beams, scores, lens = (g.cpu() for g in got)
    b, frames, _ = log_probs.shape
    for i in range(b):
        n = frames if lengths is None else min(max(int(lengths[i]), 0), frames)
        want = oracle.ctc_beam_search(log_probs[i, :n].numpy(), width, cutoff_top_n=top_n)
        for r, (tok, score) in enumerate(want):
            got_tok = beams[i, r, : int(lens[i, r])].tolist()
            # fp32 scores from two exp/log implementations: a different beam in a rank is acceptable only as a numerical tie
            assert abs(float(scores[i, r]) - score) <= 2e-4 * max(1.0, abs(score)), (i, r, float(scores[i, r]), score)
            if got_tok != tok:
                other = {tuple(t): s for t, s in want}.get(tuple(got_tok))
                assert other is not None and abs(other - score) <= 2e-4 * max(1.0, abs(score)), (i, r, got_tok, tok)
            assert torch.all(beams[i, r, int(lens[i, r]):] == 0)
        for r in range(len(want), width):
            assert int(lens[i, r]) == 0 and float(scores[i, r]) > 1e38


@pytest.mark.gpu
@pytest.mark.parametrize('b,frames,classes,width,top_n,sharp', [
    (3, 40, 49, 12, 40, 2.0), (2, 60, 49, 12, 40, 0.5), (2, 25, 5, 4, 40, 1.0), (2, 30, 49, 1, 40, 3.0), (1, 20, 49, 32, 40, 1.0),
    (2, 12, 3, 12, 40, 1.0), (1, 30, 64, 8, 10, 1.0), (2, 1, 49, 12, 40, 1.0), (1, 16, 2, 3, 1, 1.0)])
def test_gpu_beam_search_matches_oracle(b, frames, classes, width, top_n, sharp):
    from nb_asr_amd import ctc
    lp = _log_probs((b, frames, classes), 100 * frames + classes + width, sharp)
    lp[:, ::3, 0] += 1.5                                        # blanks win often, as in a trained model
    lp = torch.log_softmax(lp, dim=2)
    dev = lp.to('cuda:0')
    _check_beams(lp, None, width, top_n, ctc.beam_decode(dev, None, beam_width=width, cutoff_top_n=top_n))
    lengths = [frames, frames // 2, 0][:b]
    _check_beams(lp, lengths, width, top_n, ctc.beam_decode(dev, lengths, beam_width=width, cutoff_top_n=top_n))


@pytest.mark.gpu
def test_gpu_beam_search_narrow_beams_sweep():
    """Narrow beams over few classes: prefixes drop out of the beam and come back all the time, which is where prefix identity
    (ctcdecode's trie: a re-created prefix is the parent of its still-live extensions again) decides the result."""
    from nb_asr_amd import ctc
    rng = np.random.default_rng(7)
    for case in range(40):
        width, classes, frames = int(rng.integers(2, 5)), int(rng.integers(3, 6)), int(rng.integers(20, 61))
        lp = _log_probs((2, frames, classes), 1000 + case, sharp=float(rng.choice([0.3, 1.0, 2.0])))
        _check_beams(lp, None, width, 40, ctc.beam_decode(lp.to('cuda:0'), None, beam_width=width))


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(6))
def test_gpu_beam_search_best_beam_is_the_most_probable_labelling(seed):
    """First-principles check that needs no oracle beam search: with a beam wider than the number of labellings the best beam
    is the labelling of maximum total CTC probability (all alignments enumerated) and its score is that probability."""
    from nb_asr_amd import ctc
    rng = np.random.default_rng(seed)
    frames, classes = int(rng.integers(2, 6)), int(rng.integers(2, 4))     # at most 3 + 9 + 27 < 32 labellings of <= 3 ... 5 frames
    lp = _log_probs((1, frames, classes), 50 + seed)
    exact = oracle.ctc_labelling_log_probs(lp[0].numpy())
    if len(exact) > 32:
        pytest.skip('more labellings than the widest beam')
    beams, scores, lens = (g.cpu() for g in ctc.beam_decode(lp.to('cuda:0'), None, beam_width=32))
    best = max(exact, key=exact.get)
    assert tuple(beams[0, 0, : int(lens[0, 0])].tolist()) == best
    assert abs(-float(scores[0, 0]) - exact[best]) < 1e-4
    got = {tuple(beams[0, r, : int(lens[0, r])].tolist()): -float(scores[0, r]) for r in range(32) if float(scores[0, r]) < 1e38}
    assert set(got) == set(exact)
    for lab, v in exact.items():
        assert abs(got[lab] - v) < 1e-4


@pytest.mark.gpu
def test_gpu_beam_search_width_one_on_peaked_input_is_greedy():
    from nb_asr_amd import ctc
    gen = torch.Generator().manual_seed(5)
    path = torch.randint(0, 49, (4, 90), generator=gen)
    path[:, ::2] = 0
    logits = torch.full((4, 90, 49), -12.0).scatter_(2, path.unsqueeze(2), 6.0)
    dev = torch.log_softmax(logits, dim=2).to('cuda:0')
    beams, _, lens = ctc.beam_decode(dev, None, beam_width=12)
    greedy = ctc.greedy_decode(dev)
    for i in range(4):
        assert beams[i, 0, : int(lens[i, 0])].cpu().tolist() == greedy[i].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize('b,lh,lr,labels', [(5, 40, 30, 49), (3, 1, 1, 49), (2, 300, 120, 6), (4, 64, 65, 49), (1, 2048, 2000, 3)])
def test_gpu_token_error_counts_match_oracle(b, lh, lr, labels):
    from nb_asr_amd import ctc, hip
    gen = torch.Generator().manual_seed(lh * 7 + lr)
    hyp = torch.randint(0, labels, (b, lh), generator=gen, dtype=torch.int32)
    ref = torch.randint(0, labels, (b, lr), generator=gen, dtype=torch.int32)
    ref[:, : min(lh, lr) // 2] = hyp[:, : min(lh, lr) // 2]                  # related sequences, not just noise
    hyp_len = torch.tensor([lh, lh // 2, 0, lh, 1][:b], dtype=torch.int32)
    ref_len = torch.tensor([lr, lr, lr // 3, 0, 1][:b], dtype=torch.int32)
    dev = 'cuda:0'
    for table in (None, ctc.fold_table() if labels == 49 else torch.arange(labels, dtype=torch.int32).flip(0).clamp(max=labels - 2)):
        counts = hip.token_error_counts(hyp.to(dev), hyp_len.to(dev), ref.to(dev), ref_len.to(dev),
                                        None if table is None else table.to(dev), 0).cpu()
        want = oracle.error_counts(hyp.tolist(), hyp_len.tolist(), ref.tolist(), ref_len.tolist(), 0,
                                   None if table is None else table.tolist())
        assert [tuple(r) for r in counts.tolist()] == want
    rates = ctc.error_rates(hyp.to(dev), hyp_len, ref.to(dev), ref_len).cpu()
    want = oracle.error_counts(hyp.tolist(), hyp_len.tolist(), ref.tolist(), ref_len.tolist())
    for got, (d, n) in zip(rates.tolist(), want):
        if n:
            assert got == pytest.approx(d / n, rel=1e-6)
        else:
            assert got != got or got == float('inf')


@pytest.mark.gpu
def test_gpu_token_error_counts_reject_bad_labels():
    from nb_asr_amd import hip
    dev = 'cuda:0'
    hyp = torch.tensor([[1, 2, 77]], dtype=torch.int32, device=dev)
    ref = torch.tensor([[1, 2, 3]], dtype=torch.int32, device=dev)
    n = torch.tensor([3], dtype=torch.int32, device=dev)
    table = torch.arange(10, dtype=torch.int32, device=dev)
    assert hip.token_error_counts(hyp, n, ref, n, table, 0).cpu().tolist() == [[-1, -2]]
    long = torch.ones(1, 3000, dtype=torch.int32, device=dev)
    n_long = torch.tensor([3000], dtype=torch.int32, device=dev)
    assert hip.token_error_counts(long, n_long, ref, n, None, 0).cpu().tolist() == [[-1, -1]]


@pytest.mark.gpu
def test_gpu_decode_per_is_the_reference_decode_step():
    """Trainer.decode end to end on model output: beam search -> fold to 39 -> error rate -> mean."""
    import nb_asr_amd as nb
    from nb_asr_amd import ctc
    from nb_asr_amd.weights import keyed_fill_, keyed_input
    m = nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0)
    keyed_fill_(m, 1235, 'lively')
    m = m.to('cuda:0').eval()
    audio_len = [163, 120, 77]
    x = keyed_input(3, 163, seed=4).to('cuda:0')
    with torch.no_grad():
        log_probs = ctc.log_softmax(m(x) * 4.0)
    out_len = ctc.output_lengths(torch.tensor(audio_len))
    gen = torch.Generator().manual_seed(1)
    targets = torch.randint(1, 49, (3, 20), generator=gen, dtype=torch.int32)
    targets_len = torch.tensor([20, 11, 5], dtype=torch.int32)
    got = float(ctc.decode_per(log_probs, out_len, targets.to('cuda:0'), targets_len))
    mapping = oracle.index_mapping(_rows(), 1, 2)
    table = oracle.fold_table(mapping)
    lp = log_probs.cpu()
    hyps = [oracle.ctc_beam_search(lp[i, : int(out_len[i])].numpy(), 12)[0][0] for i in range(3)]
    width = max(len(h) for h in hyps) or 1
    hyp = [h + [0] * (width - len(h)) for h in hyps]
    counts = oracle.error_counts(hyp, [len(h) for h in hyps], targets.tolist(), targets_len.tolist(), 0, table)
    want = float(np.mean([np.float32(d) / np.float32(n) for d, n in counts]))
    assert got == pytest.approx(want, rel=1e-6)


@pytest.mark.gpu
def test_gpu_beam_search_argument_errors():
    from nb_asr_amd import hip
    lp = torch.zeros(1, 4, 65, device='cuda:0')
    with pytest.raises(hip.HipError, match='classes=65'):
        hip.ctc_beam_search(lp)
    with pytest.raises(hip.HipError, match='beam_width=33'):
        hip.ctc_beam_search(torch.zeros(1, 4, 49, device='cuda:0'), beam_width=33)
    beams, scores, lens = hip.ctc_beam_search(torch.zeros(0, 4, 49, device='cuda:0'))
    assert beams.shape == (0, 12, 4)


# ---- CTC loss value ---------------------------------------------------------------------------------------------------------

def test_oracle_ctc_loss_known_answers():
    # one frame, one label: nll = -log p(label); two frames 'a': paths (a,a), (a,_), (_,a)
    lp = torch.log(torch.tensor([[[0.25, 0.75]]]))
    assert float(oracle.ctc_loss(lp, [1], [[1]], [1])) == pytest.approx(-np.log(0.75), rel=1e-6)
    lp = torch.log(torch.tensor([[[0.4, 0.6], [0.3, 0.7]]]))
    want = -np.log(0.6 * 0.7 + 0.6 * 0.3 + 0.4 * 0.7) / 2
    assert float(oracle.ctc_loss(lp, [2], [[1]], [1])) == pytest.approx(want, rel=1e-6)
    # more labels than frames: infinite loss, zeroed (zero_infinity=True)
    assert float(oracle.ctc_loss(lp, [2], [[1, 1, 1]], [3])) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('b,frames,classes,max_labels', [(6, 40, 49, 12), (3, 250, 49, 70), (2, 7, 5, 4), (4, 130, 49, 200)])
def test_gpu_ctc_loss_matches_torch(b, frames, classes, max_labels):
    from nb_asr_amd import ctc, hip
    gen = torch.Generator().manual_seed(frames + max_labels)
    lp = _log_probs((b, frames, classes), 300 + frames, 1.5)
    targets = torch.randint(1, classes, (b, max_labels), generator=gen, dtype=torch.int32)
    targets[:, 1::3] = targets[:, 0:-1:3][:, : targets[:, 1::3].shape[1]]                 # repeated labels need a blank between them
    out_len = torch.tensor([frames, frames - 1, max(frames // 2, 1), frames, 3, frames][:b], dtype=torch.int32)
    tgt_len = torch.tensor([min(max_labels, frames // 3), 0, 1, max_labels, 2, min(max_labels, 5)][:b], dtype=torch.int32)
    dev = 'cuda:0'
    per = hip.ctc_loss(lp.to(dev), out_len.to(dev), targets.to(dev), tgt_len.to(dev), 0, True).cpu()
    want = oracle.ctc_loss(lp, out_len, targets, tgt_len, reduce=False)
    assert torch.allclose(per, want, rtol=2e-5, atol=1e-6), (per, want)
    if b > 3 and max_labels > frames:
        assert float(want[3]) == 0.0 and float(per[3]) == 0.0                              # infeasible: zeroed, like the reference
    got = float(ctc.ctc_loss(lp.to(dev), out_len, targets.to(dev), tgt_len))
    assert got == pytest.approx(float(oracle.ctc_loss(lp, out_len, targets, tgt_len)), rel=2e-5)
    raw = hip.ctc_loss(lp.to(dev), out_len.to(dev), targets.to(dev), tgt_len.to(dev), 0, False).cpu()
    assert torch.allclose(raw, want * out_len, rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_ctc_loss_edge_cases():
    from nb_asr_amd import hip
    dev = 'cuda:0'
    lp = _log_probs((3, 6, 5), 9).to(dev)
    targets = torch.tensor([[1, 2, 9], [1, 1, 1], [0, 0, 0]], dtype=torch.int32, device=dev)
    lens = torch.tensor([6, 0, 6], dtype=torch.int32, device=dev)
    tl = torch.tensor([3, 0, 0], dtype=torch.int32, device=dev)
    per = hip.ctc_loss(lp, lens, targets, tl, 0, False).cpu()
    assert per[0] != per[0]                                   # label 9 outside the 5 classes: NaN, loudly
    assert float(per[1]) == 0.0                               # no frames, no labels
    want = -float(lp[2, :, 0].sum())                          # no labels: all frames blank
    assert float(per[2]) == pytest.approx(want, rel=1e-6)
    with pytest.raises(hip.HipError, match='1024'):
        hip.ctc_loss(lp, lens, torch.zeros(3, 2000, dtype=torch.int32, device=dev), tl)


@pytest.mark.gpu
def test_gpu_evaluate_is_the_reference_validation_loop():
    """Two batches through ctc.evaluate vs the same steps assembled from the oracles (loss: ATen; decode: decode oracle)."""
    import nb_asr_amd as nb
    from nb_asr_amd import ctc
    from nb_asr_amd.weights import keyed_fill_, keyed_input
    from oracle import asr_oracle
    arch = [[1, 0], [1, 0, 0], [1, 0, 0, 0]]
    m = nb.get_model(arch, use_rnn=True, dropout_rate=0.0)
    keyed_fill_(m, 1235, 'lively')
    params = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to('cuda:0')
    gen = torch.Generator().manual_seed(3)
    table = oracle.fold_table(oracle.index_mapping(_rows(), 1, 2))
    batches, want_loss, want_per = [], [], []
    for k, (b, t) in enumerate([(3, 120), (2, 90)]):
        audio = keyed_input(b, t, seed=20 + k)
        audio_len = [t, t - 17, t // 2][:b]
        targets = torch.randint(1, 49, (b, 9), generator=gen, dtype=torch.int32)
        targets_len = torch.tensor([9, 4, 6][:b], dtype=torch.int32)
        batches.append(((audio.to('cuda:0'), audio_len), (targets.to('cuda:0'), targets_len)))
        lp = asr_oracle.log_softmax(asr_oracle.asr_forward(params, arch, audio, use_rnn=True))
        out_len = asr_oracle.output_lengths(audio_len)
        want_loss.append(float(oracle.ctc_loss(lp, out_len, targets, targets_len)))
        hyps = [oracle.ctc_beam_search(lp[i, : out_len[i]].numpy(), 12)[0][0] for i in range(b)]
        width = max(len(h) for h in hyps) or 1
        counts = oracle.error_counts([h + [0] * (width - len(h)) for h in hyps], [len(h) for h in hyps], targets.tolist(),
                                     targets_len.tolist(), 0, table)
        want_per.append(float(np.mean([np.float32(d) / np.float32(n) for d, n in counts])))
    loss, per = ctc.evaluate(m, batches)
    assert m.training                                              # restored (get_model returns a module in training mode)
    assert loss == pytest.approx(np.mean(want_loss), rel=1e-4)
    assert per == pytest.approx(np.mean(want_per), rel=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('b,frames,classes,max_labels', [(6, 40, 49, 12), (3, 250, 49, 70), (2, 7, 5, 4), (4, 130, 49, 200), (2, 30, 80, 9)])
def test_gpu_ctc_loss_gradient_matches_torch_autograd(b, frames, classes, max_labels):
    """d mean(nll / len) / d logits: the first step of the backward pass (f4), against ATen's autograd through
    log_softmax + ctc_loss -- the reference's own `loss.backward()`."""
    from nb_asr_amd import ctc
    gen = torch.Generator().manual_seed(frames * 3 + max_labels)
    logits = torch.randn(b, frames, classes, generator=gen) * 1.5
    targets = torch.randint(1, classes, (b, max_labels), generator=gen, dtype=torch.int32)
    targets[:, 1::3] = targets[:, 0:-1:3][:, : targets[:, 1::3].shape[1]]
    out_len = torch.tensor([frames, frames - 1, max(frames // 2, 1), frames, 3, frames][:b], dtype=torch.int32)
    tgt_len = torch.tensor([min(max_labels, frames // 3), 0, 1, max_labels, 2, min(max_labels, 5)][:b], dtype=torch.int32)
    want_loss, want_grad = oracle.ctc_loss_grad(logits, out_len, targets, tgt_len)
    _, truth = oracle.ctc_loss_grad(logits.double(), out_len, targets, tgt_len)          # the same call in float64
    dev = 'cuda:0'
    lp = ctc.log_softmax(logits.to(dev))
    loss, grad = ctc.ctc_loss_and_grad(lp, out_len, targets.to(dev), tgt_len)
    assert float(loss) == pytest.approx(float(want_loss), rel=2e-5)
    # the alpha-beta products of 250 frames carry fp32 rounding of ~1e-4 relative in either implementation: measure both
    # against the float64 evaluation and ask for the reference's own accuracy
    scale = float(truth.abs().max())
    err_ref = float((want_grad.double() - truth).abs().max())
    err_hip = float((grad.cpu().double() - truth).abs().max())
    assert err_hip <= max(2.0 * err_ref, 2e-6 * scale), (err_hip, err_ref, scale)
    for i in range(b):                                          # nothing flows beyond an utterance's own frames
        assert torch.all(grad[i, int(out_len[i]):] == 0)
    if b > 3 and max_labels > frames:
        assert torch.all(grad[3] == 0)                          # infeasible utterance: zero_infinity zeroes its gradient too

