"""Minimal WAV I/O and resampling for the CLI (the reference uses torchaudio.load/save with the
soundfile backend and torchaudio.functional.resample, scripts/run_optim.py:442-450, 552-565,
635-641; neither package is available here).  16/24/32-bit PCM and 32-bit float WAV via scipy;
resampling = torchaudio's windowed-sinc polyphase filter, run on the GPU (csrc/resample.hip)."""
from __future__ import annotations

import math

import numpy as np
import scipy.io.wavfile
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


_kernel_cache = {}


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample's kernel table (`_get_sinc_resample_kernel`, "sinc_interp_hann"), restated
    from the library's published algorithm (un-vendored dependency: parity unpinned): float64 design, float32
    table.  (functional.resample designs the table in the waveform's dtype, float32, and only transforms.Resample in
    float64: against the call the reference makes this table therefore agrees to float32 rounding of the design,
    about 1e-7 of the pass-band gain, not bitwise.)  -> (kernels (new, 2 * width + orig) float32, width, orig, new) with orig / new reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels *= window * scale
    return kernels.to(torch.float32), width, orig, new


def resample(audio: torch.Tensor, orig_sr: int, new_sr: int) -> torch.Tensor:
    """torchaudio.functional.resample(audio, orig_sr, new_sr) (library defaults) on the GPU through
    stito_resample_sinc; (..., n) float -> (..., ceil(n * new / orig)) on the input's device."""
    from . import _hip

    if int(orig_sr) == int(new_sr):
        return audio
    _hip.require_gpu()
    dev = audio.device if audio.is_cuda else torch.device("cuda", torch.cuda.current_device())
    key = (int(orig_sr), int(new_sr), str(dev))
    if key not in _kernel_cache:
        k, width, orig, new = sinc_resample_kernel(orig_sr, new_sr)
        _kernel_cache[key] = (k.t().contiguous().to(dev), width, orig, new)
    kt, width, orig, new = _kernel_cache[key]
    shape = audio.shape
    x = audio.detach().to(dev, torch.float32).reshape(-1, shape[-1]).contiguous()
    L = _hip.lib()
    n_out = L.stito_resample_num_samples(x.shape[1], orig, new)
    out = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=dev)
    _hip.check(L.stito_resample_sinc(_hip.ptr(x), x.shape[0], x.shape[1], _hip.ptr(kt), orig, new, width, _hip.ptr(out), n_out,
                                     _hip.stream_ptr()))
    return out.reshape(shape[:-1] + (n_out,)).to(audio.device).type_as(audio)
