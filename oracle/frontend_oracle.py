"""CPU restatement of the reference's feature front-end (TEST INFRASTRUCTURE ONLY: imported by tests/ and nothing else).

Reference: training/torch/timit.py:78-97 --
    torchaudio.transforms.MelSpectrogram(sample_rate=16000, win_length=400, hop_length=160, n_mels=80) -> torch.log ->
    (x - mean) / (variance + eps), eps = 1e-3 (timit.py:78-84), applied per utterance; batches are zero-padded in feature
    space (timit.py:54-69, 96-103).

PARITY UNPINNED: the arithmetic lives in torchaudio (not installed here, not vendored by the reference; the reference pins
no version and has no tests or golden vectors for it).  This file restates torchaudio's published algorithm with its
defaults: Spectrogram(n_fft=400, win_length=400, hop=160, window=hann_window(400) periodic, power=2, center=True,
pad_mode='reflect', normalized=False, onesided=True) = |torch.stft|^2, then MelScale(n_mels=80, sample_rate=16000, f_min=0,
f_max=8000, n_stft=201, norm=None, mel_scale='htk') = matmul with functional.melscale_fbanks.  The STFT half is anchored on
torch.stft itself (the function torchaudio calls); the filterbank half on the published formula only.
"""
import math

import numpy as np
import torch


def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(..., norm=None, mel_scale='htk'): (n_freqs, n_mels), float64."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down_slopes, up_slopes), min=0.0)


def mel_spectrogram(wave, sample_rate=16000, win_length=400, hop_length=160, n_mels=80, dtype=torch.float32):
    """wave (..., L) -> (..., n_mels, L // hop + 1), the MelSpectrogram transform at ``dtype`` precision."""
    wave = wave.to(dtype)
    n_fft = win_length
    window = torch.hann_window(win_length, periodic=True, dtype=dtype)
    lead = wave.shape[:-1]
    spec = torch.stft(wave.reshape(-1, wave.shape[-1]), n_fft=n_fft, hop_length=hop_length, win_length=win_length, window=window,
                      center=True, pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                                    # (N, n_fft/2+1, T)
    fb = melscale_fbanks_htk(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate).to(dtype)
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)          # torchaudio MelScale.forward
    return mel.reshape(*lead, n_mels, mel.shape[-1])


def features(waves, mean=None, variance=None, eps=1e-3, dtype=torch.float32, **kw):
    """List of 1-D waveforms -> zero-padded batch (B, n_mels, T_max) and the list of frame counts (timit.py collate_fn)."""
    feats = []
    for w in waves:
        f = torch.log(mel_spectrogram(w, dtype=dtype, **kw))
        if mean is not None:
            m = torch.as_tensor(np.asarray(mean), dtype=dtype)[:, None]
            v = torch.as_tensor(np.asarray(variance), dtype=dtype)[:, None]
            f = (f - m) / (v + eps)
        feats.append(f)
    t_max = max(f.shape[-1] for f in feats)
    out = torch.zeros(len(feats), feats[0].shape[0], t_max, dtype=dtype)
    for i, f in enumerate(feats):
        out[i, :, :f.shape[-1]] = f
    return out, [f.shape[-1] for f in feats]
