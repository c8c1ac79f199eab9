"""Hand-crafted audio features of the reference (st_ito/features.py), computed on the MI355X
through libstito_hip (csrc/features.hip): same function names, arguments and output shapes.

Only host-side setup lives here: the bark filterbank matrix and the FFT twiddle tables (built once
per (fft_size, sample_rate) and cached on the device, like packed weights).  compute_lufs measures on
the host exactly like the reference does (pyloudnorm there, st_ito.loudness here).
"""
from __future__ import annotations

import math
import warnings

import numpy as np
import torch

from . import _hip

_MODES = {"mono": 0, "stereo": 1, "mid-side": 2}
_cache = {}


def _hz_to_bark(freqs: float, bark_scale: str = "traunmuller") -> float:
    """reference features.py:39-67."""
    if bark_scale not in ["schroeder", "traunmuller", "wang"]:
        raise ValueError('bark_scale should be one of "schroeder", "traunmuller" or "wang".')
    if bark_scale == "wang":
        return 6.0 * math.asinh(freqs / 600.0)
    elif bark_scale == "schroeder":
        return 7.0 * math.asinh(freqs / 650.0)
    barks = ((26.81 * freqs) / (1960.0 + freqs)) - 0.53
    if barks < 2:
        barks += 0.15 * (2 - barks)
    elif barks > 20.1:
        barks += 0.22 * (barks - 20.1)
    return barks


def _bark_to_hz(barks: torch.Tensor, bark_scale: str = "traunmuller") -> torch.Tensor:
    """reference features.py:70-101, including its if / elif between the two end corrections."""
    if bark_scale not in ["schroeder", "traunmuller", "wang"]:
        raise ValueError('bark_scale should be one of "traunmuller", "schroeder" or "wang".')
    if bark_scale == "wang":
        return 600.0 * torch.sinh(barks / 6.0)
    elif bark_scale == "schroeder":
        return 650.0 * torch.sinh(barks / 7.0)
    barks = barks.clone()
    if any(barks < 2):
        idx = barks < 2
        barks[idx] = (barks[idx] - 0.3) / 0.85
    elif any(barks > 20.1):
        idx = barks > 20.1
        barks[idx] = (barks[idx] + 4.422) / 1.22
    return 1960 * ((barks + 0.53) / (26.28 - barks))


def _create_triangular_filterbank_from(all_freqs: torch.Tensor, f_pts: torch.Tensor) -> torch.Tensor:
    """Triangles between consecutive points of f_pts, (n_freqs, len(f_pts) - 2) -- reference features.py:10-36."""
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    return torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


def barkscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_barks: int, sample_rate: int,
                     bark_scale: str = "traunmuller") -> torch.Tensor:
    """Triangular bark filterbank (n_freqs, n_barks) -- reference features.py:10-36, 109-163."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_bark(f_min, bark_scale), _hz_to_bark(f_max, bark_scale), n_barks + 2)
    f_pts = _bark_to_hz(m_pts, bark_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    fb = torch.max(torch.zeros(1), torch.min((-1.0 * slopes[:, :-2]) / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
    if (fb.max(dim=0).values == 0.0).any():
        warnings.warn("At least one bark filterbank has all zero values. "
                      f"The value for `n_barks` ({n_barks}) may be set too high. "
                      f"Or, the value for `n_freqs` ({n_freqs}) may be set too low.")
    return fb


def _twiddle(n_fft: int, device) -> torch.Tensor:
    key = ("tw", n_fft, str(device))
    if key not in _cache:
        k = np.arange(n_fft // 2, dtype=np.float64)
        tw = np.stack([np.cos(-2.0 * np.pi * k / n_fft), np.sin(-2.0 * np.pi * k / n_fft)], 1).astype(np.float32)
        _cache[key] = torch.from_numpy(tw).to(device).contiguous()
    return _cache[key]


def _gpu(x: torch.Tensor):
    _hip.require_gpu()
    if x.dim() != 3:
        raise ValueError("expected (bs, chs, seq_len)")
    dev = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return x.detach().to(dev, torch.float32).contiguous(), dev


def compute_barkspectrum(x: torch.Tensor, fft_size: int = 32768, n_bands: int = 24, sample_rate: int = 44100,
                         f_min: float = 20.0, f_max: float = 20000.0, mode: str = "mid-side", **kwargs):
    """Bark spectrum embedding (bs, n_signals * n_bands), L2-normalised -- reference features.py:166-232.
    fft_size must be a power of two <= 32768 here (the transform runs in LDS)."""
    if mode not in _MODES:
        raise ValueError(f"Invalid mode {mode}")
    if fft_size & (fft_size - 1) or not 128 <= fft_size <= 32768:
        raise NotImplementedError(f"fft_size {fft_size}: only powers of two in [128, 32768] are built")
    xin, dev = _gpu(x)
    bs, chs, n = xin.shape
    key = ("fb", fft_size, n_bands, sample_rate, f_min, f_max, str(dev))
    if key not in _cache:
        _cache[key] = barkscale_fbanks(fft_size // 2 + 1, f_min, f_max, n_bands, sample_rate).T.contiguous().to(dev)
    n_sig = 1 if mode == "mono" else 2
    out = torch.empty((bs, n_sig * n_bands), dtype=torch.float32, device=dev)
    L = _hip.lib()
    _hip.check(L.stito_barkspectrum(_hip.ptr(xin), bs, chs, n, _MODES[mode], fft_size, _hip.ptr(_twiddle(fft_size, dev)),
                                    _hip.ptr(_cache[key]), n_bands, _hip.ptr(out), _hip.stream_ptr()))
    return out.to(x.device).type_as(x)


def _rms_crest(x: torch.Tensor):
    xin, dev = _gpu(x)
    bs, chs, n = xin.shape
    rms = torch.empty((bs, chs), dtype=torch.float32, device=dev)
    crest = torch.empty((bs, chs), dtype=torch.float32, device=dev)
    _hip.check(_hip.lib().stito_rms_crest(_hip.ptr(xin), bs, chs, n, _hip.ptr(rms), _hip.ptr(crest), _hip.stream_ptr()))
    return rms.to(x.device).type_as(x), crest.to(x.device).type_as(x)


def compute_rms_energy(x: torch.Tensor, **kwargs):
    """(bs, chs) -- reference features.py:235-245."""
    return _rms_crest(x)[0]


def compute_crest_factor(x: torch.Tensor, **kwargs):
    """(bs, chs) in dB -- reference features.py:248-264 (its per-sample cross-channel normalisation included)."""
    return _rms_crest(x)[1]


def compute_lufs(x: torch.Tensor, sample_rate: float, **kwargs):
    """(bs, 1) -- reference features.py:267-299: per-sample cross-channel normalisation, mono duplicated, integrated
    loudness (pyloudnorm there; the same BS.1770-4 measurement on the GPU here: stito_lufs -- K-weighting through the
    effect chain's float64 biquad cascade, gated block energies in float64; st_ito.loudness is its host restatement)."""
    from .loudness import _k_weighting

    xin, dev = _gpu(x)
    bs, chs, n = xin.shape
    sr = float(sample_rate)
    T_g, step = 0.400, 0.25
    if n < T_g * sr:
        raise ValueError("Audio must have length greater than the block size.")
    key = ("lufs", n, sr, str(dev))
    if key not in _cache:
        row = np.zeros(32)
        row[0:30:5] = 1.0  # identity sections
        for k, (b, a) in enumerate(_k_weighting(sr)):
            row[5 * k:5 * k + 5] = [b[0], b[1], b[2], a[1], a[2]]
        n_blocks = int(np.round(((n / sr - T_g) / (T_g * step))) + 1)
        lo = np.array([int(T_g * (j * step) * sr) for j in range(n_blocks)], dtype=np.int32)       # pyloudnorm's block edges
        hi = np.array([int(T_g * (j * step + 1) * sr) for j in range(n_blocks)], dtype=np.int32)
        hi = np.minimum(hi, n)
        _cache[key] = (torch.from_numpy(row).to(dev), torch.from_numpy(lo).to(dev), torch.from_numpy(hi).to(dev), n_blocks)
    row, lo, hi, n_blocks = _cache[key]
    coef = row[None, :].repeat(bs, 1).contiguous()
    L = _hip.lib()
    ws = torch.empty(L.stito_lufs_workspace_bytes(bs, n, n_blocks), dtype=torch.uint8, device=dev)
    out = torch.empty((bs, 1), dtype=torch.float32, device=dev)
    _hip.check(L.stito_lufs(_hip.ptr(xin), bs, chs, n, _hip.ptr(coef), _hip.ptr(lo), _hip.ptr(hi), n_blocks, 1.0 / (T_g * sr),
                            _hip.ptr(out), _hip.ptr(ws), ws.numel(), _hip.stream_ptr()))
    return out.to(x.device).type_as(x)


def compute_spectral_centroid(x: torch.Tensor, sample_rate: float, *args, **kwargs):
    """(bs, chs * 10) -- reference features.py:302-333 (torchaudio SpectralCentroid, n_fft 2048, hop 1024)."""
    xin, dev = _gpu(x)
    bs, chs, n = xin.shape
    key = ("hann", 2048, str(dev))
    if key not in _cache:
        _cache[key] = torch.hann_window(2048, periodic=True, dtype=torch.float64).to(torch.float32).to(dev)
    L = _hip.lib()
    ws = torch.empty(L.stito_spectral_centroid_workspace_bytes(bs, chs, n), dtype=torch.uint8, device=dev)
    out = torch.empty((bs, chs * 10), dtype=torch.float32, device=dev)
    _hip.check(L.stito_spectral_centroid(_hip.ptr(xin), bs, chs, n, float(sample_rate), _hip.ptr(_cache[key]),
                                         _hip.ptr(_twiddle(2048, dev)), _hip.ptr(out), _hip.ptr(ws), ws.numel(), _hip.stream_ptr()))
    return out.to(x.device).type_as(x)
