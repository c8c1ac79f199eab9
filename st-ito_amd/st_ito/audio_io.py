"""Minimal WAV I/O and resampling for the CLI (the reference uses torchaudio.load/save with the
soundfile backend and torchaudio.functional.resample, scripts/run_optim.py:442-450, 552-565,
635-641; neither package is available here).  16/24/32-bit PCM and 32-bit float WAV via scipy."""
from __future__ import annotations

from fractions import Fraction

import numpy as np
import scipy.io.wavfile
import scipy.signal
import torch


def load_wav(path: str):
    """-> (audio (chs, n) float32 in [-1, 1], sample_rate)."""
    sr, data = scipy.io.wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def save_wav(path: str, audio: torch.Tensor, sample_rate: int):
    """audio (chs, n) float -> 32-bit float WAV."""
    a = audio.detach().cpu().to(torch.float32).numpy()
    if a.ndim == 1:
        a = a[None]
    scipy.io.wavfile.write(path, int(sample_rate), np.ascontiguousarray(a.T))


def resample(audio: torch.Tensor, orig_sr: int, new_sr: int) -> torch.Tensor:
    """Polyphase resampling (stands in for torchaudio.functional.resample; the two differ in
    their anti-aliasing filter, so resampled audio is not bit-compatible with the reference)."""
    if orig_sr == new_sr:
        return audio
    fr = Fraction(int(new_sr), int(orig_sr))
    y = scipy.signal.resample_poly(audio.numpy().astype(np.float64), fr.numerator, fr.denominator, axis=-1)
    return torch.from_numpy(y.astype(np.float32))
