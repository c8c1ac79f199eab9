"""Integrated loudness (ITU-R BS.1770-4) -- what the reference's evaluation harness gets from
`pyloudnorm.Meter(sr).integrated_loudness(x)` to bring outputs and targets to -22 LUFS before they
are saved (scripts/eval/eval_pst.py:846-853, 877-885).  pyloudnorm is an un-vendored dependency
(setup.py) absent from this image; this is the published algorithm: K-weighting (high-shelf
+4 dB at 1.5 kHz, high-pass at 38 Hz), 400 ms blocks with 75 % overlap, absolute gate -70 LUFS,
relative gate -10 LU.  Post-processing of a handful of files on the host: numpy/scipy.
"""
from __future__ import annotations

import numpy as np
import scipy.signal


def _k_weighting(sr: float):
    """The two biquads of pyloudnorm's default "K-weighting" filter class (RBJ-style designs)."""
    def shelf(G, Q, fc):
        A = 10 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / sr)
        alpha = np.sin(w0) / (2.0 * Q)
        b = np.array([A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha),
                      -2 * A * ((A - 1) + (A + 1) * np.cos(w0)),
                      A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)])
        a = np.array([(A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha,
                      2 * ((A - 1) - (A + 1) * np.cos(w0)),
                      (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha])
        return b / a[0], a / a[0]

    def highpass(Q, fc):
        w0 = 2.0 * np.pi * (fc / sr)
        alpha = np.sin(w0) / (2.0 * Q)
        b = np.array([(1 + np.cos(w0)) / 2, -(1 + np.cos(w0)), (1 + np.cos(w0)) / 2])
        a = np.array([1 + alpha, -2 * np.cos(w0), 1 - alpha])
        return b / a[0], a / a[0]

    return [shelf(4.0, 1 / np.sqrt(2), 1500.0), highpass(0.5, 38.0)]


def integrated_loudness(data: np.ndarray, sr: float, block_size: float = 0.400) -> float:
    """data: (samples, channels) or (samples,), like pyloudnorm.  Returns LUFS (-inf for silence)."""
    x = np.asarray(data, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    n, chs = x.shape
    if chs > 5:
        raise ValueError("Audio must have five channels or less.")
    if n < block_size * sr:
        raise ValueError("Audio must have length greater than the block size.")
    for b, a in _k_weighting(sr):
        x = scipy.signal.lfilter(b, a, x, axis=0)
    G = [1.0, 1.0, 1.0, 1.41, 1.41]
    T_g, overlap, gamma_a = block_size, 0.75, -70.0
    step = 1.0 - overlap
    T = n / sr
    n_blocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
    z = np.zeros((chs, n_blocks))
    for i in range(chs):
        for j in range(n_blocks):
            lo = int(T_g * (j * step) * sr)
            hi = int(T_g * (j * step + 1) * sr)
            z[i, j] = (1.0 / (T_g * sr)) * np.sum(np.square(x[lo:hi, i]))
    with np.errstate(divide="ignore"):
        l = np.array([-0.691 + 10.0 * np.log10(np.sum([G[i] * z[i, j] for i in range(chs)])) for j in range(n_blocks)])
    J_g = [j for j, lj in enumerate(l) if lj >= gamma_a]
    if not J_g:
        return -np.inf
    z_avg = [np.mean([z[i, j] for j in J_g]) for i in range(chs)]
    gamma_r = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg[i] for i in range(chs)])) - 10.0
    J_g = [j for j, lj in enumerate(l) if lj > gamma_r and lj > gamma_a]
    if not J_g:
        return -np.inf
    z_avg = [np.mean([z[i, j] for j in J_g]) for i in range(chs)]
    with np.errstate(divide="ignore"):
        return float(-0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg[i] for i in range(chs)])))


def normalize_loudness(audio, sr: float, target_lufs: float = -22.0):
    """audio (chs, n) torch tensor or array -> scaled to target_lufs (eval_pst.py:846-853)."""
    import torch

    arr = audio.detach().cpu().numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
    lufs = integrated_loudness(arr.T, sr)
    gain_lin = 10 ** ((target_lufs - lufs) / 20)
    return audio * gain_lin, lufs
