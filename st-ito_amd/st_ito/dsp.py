"""Random-parameter effect wrappers of the reference's st_ito/dsp.py (used there to fabricate
"styles" for datasets; nothing on the ES path calls them) on the MI355X renderer.

  apply_random_simple_distortion  dsp.py:12-23   tanh(x * 10^(drive_db/20)), drive_db ~ U(0, 32)
  apply_random_reverb             dsp.py:26-46   dasp noise_shaped_reverberation, fixed band decays/gains,
                                                 mix ~ U(0, 1)  -> csrc/convreverb.hip
  apply_random_compressor         dsp.py:49-78   dasp_pytorch.functional.compressor (un-vendored: restated, parity
                                                 unpinned): soft-knee gain computer + the library's single
                                                 one-pole smoothing filter, which it applies by frequency sampling
                                                 = the causal recursion to FFT rounding  -> csrc/modfx.hip
  normalize_loudness              dsp.py:81-87   BS.1770 integrated loudness on the host (st_ito.loudness)

Inputs/outputs are (chs, n) or (bs, chs, n) tensors like dasp's (the reference passes what its
dataset code holds); `use_gpu` is accepted and ignored (the renderer always runs on the GPU).
"""
from __future__ import annotations

import numpy as np
import torch

from . import effects as _fx

_REVERB = {}


def _render(instance, x: torch.Tensor, sample_rate: float) -> torch.Tensor:
    batched = x.dim() == 3
    items = x if batched else x[None]
    out = [torch.from_numpy(instance.process(it.detach().cpu().numpy(), sample_rate)) for it in items]
    y = torch.stack(out)
    return y if batched else y[0]


def apply_random_simple_distortion(x: torch.Tensor, sample_rate: float, use_gpu: bool = False):
    drive_db = np.random.uniform(0, 32)
    fx = _fx.BasicDistortion()
    fx.parameters["drive_db"].set_value(float(np.float32(drive_db)))
    fx.parameters["output_gain_db"].set_value(0.0)
    return _render(fx, x, sample_rate), drive_db


def apply_random_reverb(x: torch.Tensor, sample_rate: float, use_gpu: bool = False, seed: int = 0):
    """The library draws a fresh noise impulse response per call; here the noise bank is seeded
    (`seed`), so a call is reproducible given numpy's RNG state for `mix`."""
    mix = np.random.uniform(0, 1.0)
    key = (int(sample_rate), seed)
    if key not in _REVERB:
        _REVERB[key] = _fx.NoiseShapedReverb(sample_rate=sample_rate, seed=seed)
    fx = _REVERB[key]
    for b, d in enumerate([0.6, 0.4, 0.4, 0.5, 0.2, 0.3, 0.3, 0.2, 0.1, 0.1, 0.2, 0.1]):
        fx.parameters[f"band{b}_decay"].set_value(float(np.float32(d)))
        fx.parameters[f"band{b}_gain"].set_value(1.0)
    fx.parameters["mix"].set_value(float(np.float32(mix)))
    return _render(fx, x, sample_rate).cpu(), mix


def apply_random_compressor(x: torch.Tensor, sample_rate: float, use_gpu: bool = False):
    """threshold ~ U(-48, 0) dB, ratio 4, attack = release = 100 ms, knee 24 dB, no make-up (dsp.py:49-78).  x (bs, chs, n) like
    dasp's compressor (a (chs, n) tensor is taken as one item)."""
    from . import _hip

    threshold_db = np.random.uniform(-48, 0)
    _hip.require_gpu()
    batched = x.dim() == 3
    dev = torch.device("cuda", torch.cuda.current_device())
    xin = (x if batched else x[None]).detach().to(dev, torch.float32).contiguous()
    bs, chs, n = xin.shape
    out = torch.empty_like(xin)
    _hip.check(_hip.lib().stito_dasp_compressor(_hip.ptr(xin), bs, chs, n, float(sample_rate), float(np.float32(threshold_db)), 4.0, 100.0,
                                                24.0, 0.0, _hip.ptr(out), _hip.stream_ptr()))
    y = out.cpu()
    return (y if batched else y[0]), threshold_db


def normalize_loudness(x: torch.Tensor, sr: float, target_lufs_db: float):
    from .loudness import integrated_loudness

    x_lufs_db = integrated_loudness(x.permute(1, 0).numpy(), sr)
    delta_lufs_db = torch.tensor([target_lufs_db - x_lufs_db]).float()
    gain_lin = 10.0 ** (delta_lufs_db.clamp(-120, 40.0) / 20.0)
    return gain_lin * x
