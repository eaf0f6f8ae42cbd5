"""Post-logits step: log_softmax + length mapping + greedy CTC decode."""
import pytest
import torch

from oracle import asr_oracle as oracle


def test_oracle_greedy_hand_cases():
    def onehot(seq, classes=5):
        x = torch.full((1, len(seq), classes), -5.0)
        for t, c in enumerate(seq):
            x[0, t, c] = 3.0
        return x
    assert oracle.ctc_greedy(onehot([0, 1, 1, 0, 1, 2, 2, 0, 0, 3])) == [[1, 1, 2, 3]]
    assert oracle.ctc_greedy(onehot([0, 0, 0])) == [[]]
    assert oracle.ctc_greedy(onehot([2, 2, 2, 2])) == [[2]]
    assert oracle.ctc_greedy(onehot([1, 2, 3, 4]), lengths=[2]) == [[1, 2]]
    assert oracle.ctc_greedy(onehot([1, 0, 1]), blank=1) == [[0]]
    assert oracle.output_lengths([1000, 1001, 1003, 3, 4]) == [250, 250, 250, 0, 1]         # trainer.py:219 floor rule
    x = torch.randn(2, 7, 49)
    assert torch.allclose(oracle.log_softmax(x).exp().sum(dim=2), torch.ones(2, 7), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('b,t,c', [(1, 1, 49), (3, 250, 49), (2, 257, 49), (4, 700, 49), (2, 40, 5)])
def test_gpu_postprocess_matches_oracle(b, t, c):
    from nb_asr_amd import ctc, hip
    torch.manual_seed(b * 1000 + t)
    logits = torch.randn(b, t, c) * 2.0
    logits[:, ::3, 0] += 4.0                                   # plenty of blanks
    logits[:, 1::5] = logits[:, 0:-1:5][:, : logits[:, 1::5].shape[1]]     # and repeated frames
    dev = logits.to('cuda:0')
    got_lp = ctc.log_softmax(dev).cpu()
    want_lp = oracle.log_softmax(logits)
    assert float((got_lp - want_lp).abs().max()) < 2e-6
    audio_len = [4 * t, 4 * t - 1, 2 * t, 7][:b]
    seqs, lp2 = ctc.greedy_decode(dev, audio_len, return_log_probs=True)
    want = oracle.ctc_greedy(logits, oracle.output_lengths(audio_len))
    assert [s.tolist() for s in seqs] == want
    assert torch.equal(lp2.cpu(), got_lp)
    assert [s.tolist() for s in ctc.greedy_decode(dev)] == oracle.ctc_greedy(logits)
    _, tokens, counts = hip.ctc_postprocess(dev, None, False, True)
    for i in range(b):
        assert torch.all(tokens[i, int(counts[i]):] == -1)


@pytest.mark.gpu
def test_gpu_postprocess_on_model_output():
    import nb_asr_amd as nb
    from nb_asr_amd import ctc
    from nb_asr_amd.weights import keyed_fill_, keyed_input
    arch = [[3, 1], [4, 1, 1], [2, 1, 1, 1]]
    m = nb.get_model(arch, use_rnn=True, dropout_rate=0.0)
    keyed_fill_(m, 1235, 'lively')
    m = m.to('cuda:0').eval()
    x = keyed_input(3, 203, seed=2).to('cuda:0')
    with torch.no_grad():
        logits = m(x)
    seqs = ctc.greedy_decode(logits, [203, 150, 99])
    want = oracle.ctc_greedy(logits.cpu(), oracle.output_lengths([203, 150, 99]))
    assert [s.tolist() for s in seqs] == want
