"""Feature front-end: host-side matrices against the oracle on CPU, the HIP chain against the oracle
on the GPU.  The oracle's own anchor is torch.stft (oracle/frontend_oracle.py explains what is and is not pinned)."""
import math
import pathlib

import numpy as np
import pytest
import torch

from nb_asr_amd import frontend
from oracle import frontend_oracle as fo

GOLDEN = pathlib.Path(__file__).resolve().parent / 'golden' / 'frontend_fixtures.npz'
DEV = 'cuda:0'


def keyed_wave(seed, samples):
    rng = np.random.default_rng(seed)
    t = np.arange(samples) / 16000.0
    tone = 0.3 * np.sin(2 * math.pi * (200.0 + 37.0 * seed) * t) + 0.2 * np.sin(2 * math.pi * 3100.0 * t)
    return torch.from_numpy((tone + 0.1 * rng.standard_normal(samples)).astype(np.float32))


# ---- CPU: host logic and the oracle ------------------------------------------------------------------------------------
def test_mel_filterbank_matches_the_published_formula():
    fb = frontend.mel_filterbank(201, 80, 16000)
    want = fo.melscale_fbanks_htk(201, 0.0, 8000.0, 80, 16000).numpy()
    assert fb.shape == (201, 80)
    np.testing.assert_allclose(fb, want, rtol=0, atol=1e-12)
    assert (fb >= 0).all() and (fb.max(axis=0) > 0.2).all()          # no empty filter at 201 bins (torchaudio warns when one is)
    centres = fb.argmax(axis=0)
    assert (np.diff(centres) >= 0).all() and centres[0] >= 1 and centres[-1] <= 199
    # HTK mel scale known answers: 1000 Hz <-> 999.99 mel
    assert abs(float(frontend.hz_to_mel_htk(1000.0)) - 999.9855) < 1e-3
    assert abs(float(frontend.mel_to_hz_htk(frontend.hz_to_mel_htk(4321.0))) - 4321.0) < 1e-9


def test_windowed_dft_matrix_reproduces_torch_stft():
    wave = keyed_wave(1, 4000).double()
    m = frontend.windowed_dft_matrix(400, 400)                       # (402, 400)
    assert m.shape == (402, 400)
    spec = torch.stft(wave, n_fft=400, hop_length=160, win_length=400, window=torch.hann_window(400, periodic=True, dtype=torch.float64),
                      center=True, pad_mode='reflect', return_complex=True)          # (201, 26)
    padded = torch.nn.functional.pad(wave[None, None], (200, 200), mode='reflect')[0, 0].numpy()
    frames = np.stack([padded[t * 160:t * 160 + 400] for t in range(spec.shape[1])], axis=1)     # (400, T)
    y = m @ frames
    np.testing.assert_allclose(y[:201], spec.real.numpy(), atol=1e-10)
    np.testing.assert_allclose(y[201:], spec.imag.numpy(), atol=1e-10)


def test_oracle_frame_count_padding_and_normalisation():
    stats = np.load(GOLDEN)
    waves = [keyed_wave(2, 1600), keyed_wave(3, 1000), keyed_wave(4, 1759)]
    feats, frames = fo.features(waves, stats['moving_mean'], stats['moving_variance'])
    assert frames == [11, 7, 11] and tuple(feats.shape) == (3, 80, 11)
    assert torch.all(feats[1, :, 7:] == 0)                          # zero padding in FEATURE space (timit.py:54-69)
    raw, _ = fo.features(waves[:1])
    want = (raw[0] - torch.from_numpy(stats['moving_mean'])[:, None]) / (torch.from_numpy(stats['moving_variance'])[:, None] + 1e-3)
    torch.testing.assert_close(feats[0], want)                      # divides by the variance, not its root (timit.py:83)


def test_frontend_refuses_cpu_tensors():
    from nb_asr_amd import hip
    with pytest.raises((hip.HipError, RuntimeError, AssertionError)):
        frontend.LogMelFrontend(device='cpu')(torch.zeros(1, 1600))


# ---- GPU: the HIP chain ----------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('lengths', [[16000, 16000], [16000, 9999, 12345], [1600], [401, 3000]])
def test_frontend_matches_oracle(lengths):
    stats = np.load(GOLDEN)
    waves = [keyed_wave(10 + i, n) for i, n in enumerate(lengths)]
    want, frames = fo.features(waves, stats['moving_mean'], stats['moving_variance'])
    truth, _ = fo.features(waves, stats['moving_mean'], stats['moving_variance'], dtype=torch.float64)
    batch = torch.zeros(len(waves), max(lengths))
    for i, w in enumerate(waves):
        batch[i, :w.numel()] = w
    fe = frontend.LogMelFrontend(mean=stats['moving_mean'], variance=stats['moving_variance'], device=DEV)
    got = fe(batch.to(DEV), lengths if len(set(lengths)) > 1 else None).cpu()
    assert tuple(got.shape) == tuple(want.shape) and torch.isfinite(got).all()
    for i, t in enumerate(frames):
        assert torch.all(got[i, :, t:] == 0)
    # fp32 log-mel of an energetic signal: both the HIP chain and the fp32 oracle sit within a few 1e-5 of the fp64 evaluation
    err_hip = float((got.double() - truth).abs().max())
    err_cpu = float((want.double() - truth).abs().max())
    assert err_hip <= max(3.0 * err_cpu, 2e-5), (err_hip, err_cpu)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_frontend_feeds_the_model():
    import nb_asr_amd as nb
    from nb_asr_amd.weights import keyed_fill_
    fe = frontend.LogMelFrontend(device=DEV)
    feats = fe(torch.stack([keyed_wave(20, 8000), keyed_wave(21, 8000)]).to(DEV))
    assert tuple(feats.shape) == (2, 80, 51)
    model = keyed_fill_(nb.get_model([[1, 0], [1, 0, 0], [1, 0, 0, 0]], use_rnn=True, dropout_rate=0.0), mode='lively').to(DEV).eval()
    with torch.no_grad():
        logits = model(feats)
    assert tuple(logits.shape) == (2, 13, 49) and torch.isfinite(logits).all()


@pytest.mark.gpu
def test_pointwise_linear_vs_torch():
    from nb_asr_amd import hip
    torch.manual_seed(0)
    x = torch.randn(2, 36, 52, device=DEV)
    x[:, :, 50:] = 0
    w, bias = torch.randn(21, 36, device=DEV), torch.randn(21, device=DEV)
    y = torch.full((2, 21, 52), float('nan'), device=DEV)
    hip.pointwise_linear(x, 50, w, bias, y)
    want = torch.einsum('oc,bct->bot', w.double().cpu(), x.double().cpu()[:, :, :50]) + bias.double().cpu()[None, :, None]
    assert torch.all(y[:, :, 50:] == 0)
    torch.testing.assert_close(y[:, :, :50].cpu().double(), want, rtol=1e-5, atol=1e-5)
