"""Feature front-end on the HIP device: waveform -> normalised log-mel features in the model's input layout.

Mirrors the transform chain of the reference's TIMIT loader (reference training/torch/timit.py:78-97):

    torchaudio.transforms.MelSpectrogram(sample_rate=16000, win_length=400, hop_length=160, n_mels=80)
    torch.log
    (x - mean) / (variance + eps)        # NB: the variance, not its square root (timit.py:83), eps = 1e-3

with torchaudio's defaults spelled out: n_fft = 400, centred frames with reflect padding, periodic Hann window, power 2,
HTK mel scale between 0 and sample_rate / 2, no filterbank normalisation.  The reference applies the chain per utterance and
zero-pads the FEATURES of a batch (timit.py:54-69, 96-103); ``lengths`` reproduces that for a zero-padded batch of waveforms.

All arithmetic runs in libnbasr_hip.so (frontend.hip + the fp32 MFMA GEMM); there is no CPU path.
"""
import math

import numpy as np
import torch

from . import hip


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(n_freqs, n_mels, sample_rate, f_min=0.0, f_max=None):
    """(n_freqs, n_mels) triangular HTK filterbank, torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk')."""
    f_max = sample_rate / 2.0 if f_max is None else f_max
    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    f_pts = mel_to_hz_htk(np.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2))
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def windowed_dft_matrix(n_fft, win_length):
    """(2 * (n_fft/2 + 1), n_fft): rows 0..n_fft/2 = w[n] cos(2 pi k n / N), then the same with -sin (periodic Hann w)."""
    n = np.arange(n_fft, dtype=np.float64)
    window = np.zeros(n_fft)
    left = (n_fft - win_length) // 2
    window[left:left + win_length] = 0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(win_length) / win_length)
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * math.pi * k * n[None, :] / n_fft
    return np.concatenate([np.cos(ang) * window, -np.sin(ang) * window], axis=0)


class LogMelFrontend:
    """``frontend(wave, lengths=None) -> (B, n_mels, T)`` float32 on ``wave``'s HIP device, ``T = L // hop + 1``.

    ``wave``: (B, L) float32 device tensor (zero-padded batch); ``lengths``: per-utterance sample counts (sequence or int32
    tensor) or None.  ``mean`` / ``variance``: per-mel statistics (the reference's ``timit_train_stats.npz`` arrays
    ``moving_mean`` / ``moving_variance``); None = no normalisation (plain log-mel)."""

    def __init__(self, sample_rate=16000, win_length=400, hop_length=160, n_mels=80, n_fft=None, mean=None, variance=None,
                 eps=1e-3, device='cuda:0'):
        self.sample_rate, self.win_length, self.hop_length, self.n_mels = sample_rate, win_length, hop_length, n_mels
        self.n_fft = win_length if n_fft is None else n_fft
        if self.n_fft % 4:
            raise ValueError('n_fft must be a multiple of 4 (GEMM K alignment)')
        self.device = torch.device(device)
        self.bins = self.n_fft // 2 + 1
        self.bins_padded = (self.bins + 3) // 4 * 4
        dft = windowed_dft_matrix(self.n_fft, win_length)
        fb = np.zeros((n_mels, self.bins_padded))
        fb[:, :self.bins] = mel_filterbank(self.bins, n_mels, sample_rate).T
        mean = np.zeros(n_mels) if mean is None else np.asarray(mean, dtype=np.float64)
        inv = np.ones(n_mels) if variance is None else 1.0 / (np.asarray(variance, dtype=np.float64) + eps)
        if mean.shape != (n_mels,) or inv.shape != (n_mels,):
            raise ValueError(f'mean and variance must have {n_mels} entries')
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)
        self.dft, self.fbank, self.mean, self.inv_scale = f32(dft), f32(fb), f32(mean), f32(inv)
        self.zero_bias = torch.zeros(max(2 * self.bins, n_mels), device=self.device)

    def num_frames(self, samples):
        return samples // self.hop_length + 1 if samples > 0 else 0

    def __call__(self, wave, lengths=None):
        if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_cuda:
            raise hip.HipError('wave must be a (batch, samples) float32 tensor on a HIP device; this package has no CPU path')
        wave = wave.contiguous()
        b, samples = wave.shape
        if lengths is not None and not torch.is_tensor(lengths):
            lengths = torch.tensor(list(lengths), dtype=torch.int32)
        if lengths is not None:
            if int(lengths.max()) > samples or int(lengths.min()) <= self.n_fft // 2:
                raise ValueError(f'lengths must lie in ({self.n_fft // 2}, {samples}]')
            lengths = lengths.to(device=wave.device, dtype=torch.int32).contiguous()
        t = self.num_frames(samples)
        ld = hip.round_up4(t)
        frames = torch.empty(b, self.n_fft, ld, device=wave.device)
        hip.frame_signal(wave, lengths, frames, self.n_fft, self.hop_length)
        spec = torch.empty(b, 2 * self.bins, ld, device=wave.device)
        hip.pointwise_linear(frames, t, self.dft, self.zero_bias, spec)
        power = torch.empty(b, self.bins_padded, ld, device=wave.device)
        hip.power_spectrum(spec, self.bins, power)
        mel = torch.empty(b, self.n_mels, ld, device=wave.device)
        hip.pointwise_linear(power, t, self.fbank, self.zero_bias, mel)
        hip.log_normalize(mel, lengths, self.mean, self.inv_scale, mel, samples, self.hop_length)
        return mel[:, :, :t] if ld != t else mel
