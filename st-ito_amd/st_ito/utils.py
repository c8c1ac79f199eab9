"""Loader / embedding-function pair of the AFx-Rep metric: the reference's
st_ito/utils.py:444-551 (`get_param_embeds`, `load_param_model`) plus `apply_fade_in` (31-43).
The other metrics of the reference's utils.py (CLAP, BEATs, wav2vec2, ...) are outside this
build's scope."""
from __future__ import annotations

import math
import os
from importlib import import_module

import numpy as np
import torch
import yaml

from . import _hip


def apply_fade_in(x: torch.Tensor, num_samples: int = 16384):
    """reference utils.py:31-43."""
    fade = torch.linspace(0, 1, num_samples, device=x.device)
    x[..., :num_samples] = x[..., :num_samples] * fade
    return x


def get_param_embeds(
    x: torch.Tensor,
    model: torch.nn.Module,
    sample_rate: float,
    requires_grad: bool = False,
    peak_normalize: bool = False,
    dropout: float = 0.0,
):
    """reference utils.py:444-508.  x: (bs, chs, seq_len) on any device -> {"mid","side"}: (bs, E)
    L2-normalised, returned with x's device/dtype.

    The per-item peak normalisation (utils.py:473-474) is fused into the STFT loader of the HIP
    front-end; the NaN scrub and F.normalize run in stito_embed_loss.

    Side effect kept from the reference: there `x.type_as(model parameter)` returns x itself when x already has the
    model's dtype and device type and no resampling happens, so lines 473-474 peak-normalise the CALLER's tensor in
    place.  The HIP model always lives on the GPU; the device the reference's model would be on is the one
    load_param_model was asked for (`model.reference_device`: "cuda" for use_gpu=True, else "cpu"; a model built
    directly counts as living where its parameters are).  When x is float32 on that device type at 48 kHz it comes back
    normalised, exactly as from the reference; otherwise it is left untouched, as there.  One deviation: an expanded
    (stride-0) view, on which the reference raises, is left un-normalised and still gets its embeddings."""
    if x.dim() != 3:
        raise ValueError("expected (bs, chs, seq_len)")
    if requires_grad:
        raise NotImplementedError("get_param_embeds(requires_grad=True) (the autodiff path) is not built here")
    _hip.require_gpu()
    from .models.panns import Cnn14

    if not isinstance(model, Cnn14):
        raise TypeError("get_param_embeds (MI355X build) needs the Cnn14 returned by load_param_model")
    x_device = x
    dev = next(model.parameters()).device
    xin = x.detach().to(dev, torch.float32).contiguous()
    if sample_rate != 48000:  # utils.py:462-463
        from .audio_io import resample

        xin = resample(xin, int(sample_rate), 48000).contiguous()
    bs, chs, n = xin.shape
    L = _hip.lib()
    st = _hip.stream_ptr()
    peaks = torch.empty(bs, dtype=torch.float32, device=dev)
    _hip.check(L.stito_peak(_hip.ptr(xin), bs, chs, n, _hip.ptr(peaks), st))
    mid, side = model.embed_raw(xin, peaks, norm_passes=1)
    if dropout > 0.0:
        mid = torch.nn.functional.dropout(mid, p=dropout, training=True)
        side = torch.nn.functional.dropout(side, p=dropout, training=True)
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    _hip.check(L.stito_embed_loss(_hip.ptr(mid), _hip.ptr(side), bs, mid.shape[1], None, None, None, _hip.ptr(flags), st))
    fl = flags.cpu()
    if fl[0]:
        print("Warning: NaNs found in mid_embeddings")
    elif fl[1]:
        print("Warning: NaNs found in side_embeddings")
    ref_dev = getattr(model, "reference_device", None) or dev.type
    if x.dtype == torch.float32 and sample_rate == 48000 and x.device.type == ref_dev and not x.requires_grad:
        # utils.py:473-474 on the caller's tensor.  DEVIATION: for an expanded view (a stride-0 dimension: several elements
        # share one memory location) the reference's in-place division raises and its embeddings are lost; here that one
        # case is detected up front, the view is left untouched and the embeddings are returned.  Any other failure of
        # the division (a HIP error, a read-only tensor) propagates.
        shared = any(st == 0 and sz > 1 for st, sz in zip(x.stride(), x.shape))
        if not shared:
            x.div_(peaks.to(x.device).clamp(min=1e-8).view(-1, 1, 1))
    return {"mid": mid.type_as(x_device), "side": side.type_as(x_device)}


def _load_checkpoint(ckpt_path: str) -> dict:
    """torch.load of a (downloaded) Lightning checkpoint WITHOUT executing pickled code.

    The reference calls torch.load(ckpt_path, map_location="cpu") (utils.py:536), which on the torch it
    was written for unpickles arbitrary objects.  Here the file is read with weights_only=True; classes
    the checkpoint mentions outside torch's allow-list (Lightning / jsonargparse hyper-parameter
    containers, callbacks state, ...) are mapped to inert stand-ins of the same qualified name -- their
    constructor arguments and state are discarded, no constructor or reducer of the real class ever runs -- since
    only checkpoint["state_dict"] is used.  STITO_TRUST_CHECKPOINT=1 restores the reference's full unpickle."""
    import pickle
    import re

    if os.environ.get("STITO_TRUST_CHECKPOINT", "0") == "1":
        return torch.load(ckpt_path, map_location="cpu", weights_only=False)
    stand_ins = []
    for _ in range(64):
        try:
            with torch.serialization.safe_globals(stand_ins):
                return torch.load(ckpt_path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError as e:
            m = re.search(r"Unsupported global: GLOBAL ([\w\.]+) was not an allowed global", str(e))
            if m is None or any(f"{c.__module__}.{c.__qualname__}" == m.group(1) for c in stand_ins):
                raise
            module, _, name = m.group(1).rpartition(".")
            stand_ins.append(type(name, (), {"__module__": module, "__qualname__": name,
                                             "__new__": lambda cls, *a, **k: object.__new__(cls),
                                             "__init__": lambda self, *a, **k: None,
                                             "__setstate__": lambda self, state: None,
                                             "__call__": lambda self, *a, **k: None}))
        except Exception as e:  # a foreign object the restricted unpickler cannot rebuild even as an inert stand-in
            raise pickle.UnpicklingError(
                f"{ckpt_path}: cannot be read without executing pickled code ({type(e).__name__}: {e}). If the file is "
                "trusted, set STITO_TRUST_CHECKPOINT=1 to load it with the reference's full torch.load") from e
    raise pickle.UnpicklingError(f"{ckpt_path}: too many foreign classes in the checkpoint")


def load_param_model(ckpt_path: str = None, use_gpu: bool = False):
    """reference utils.py:511-551.  Reads config.yaml next to the checkpoint, builds the encoder
    (class path `lcap.*`/`st_ito.*` -> this package), loads the `encoder.*` weights strictly.

    There is no network here: the reference's wget of afx-rep.ckpt is not attempted.  The HIP
    forward only exists on the GPU, so the model is moved to the current HIP device whenever one
    is visible (use_gpu=False then only changes where get_param_embeds returns its result,
    exactly like the reference: embeddings come back on the input's device)."""
    if ckpt_path is None:
        ckpt_path = os.path.join(os.getcwd(), "tmp", "afx-rep.ckpt")
    if not os.path.isfile(ckpt_path):
        raise FileNotFoundError(
            f"{ckpt_path} not found.  Download afx-rep.ckpt and config.yaml from "
            "https://huggingface.co/csteinmetz1/afx-rep into that directory (no network access here), "
            "or use st_ito.utils.make_synthetic_param_model() for benchmarking.")
    config_path = os.path.join(os.path.dirname(ckpt_path), "config.yaml")
    with open(config_path) as f:
        config = yaml.safe_load(f)
    encoder_configs = config["model"]["init_args"]["encoder"]
    module_path, class_name = encoder_configs["class_path"].rsplit(".", 1)
    module_path = module_path.replace("lcap", "st_ito")
    module = import_module(module_path)
    model = getattr(module, class_name)(**encoder_configs["init_args"])
    checkpoint = _load_checkpoint(ckpt_path)
    state_dict = {}
    for k, v in checkpoint["state_dict"].items():
        if k.startswith("encoder"):
            state_dict[k.replace("encoder.", "", 1)] = v
    model.load_state_dict(state_dict)
    model.eval()
    if use_gpu or torch.cuda.is_available():
        model.cuda()
    model.reference_device = "cuda" if use_gpu else "cpu"  # where the reference's model would live (get_param_embeds)
    return model


def make_synthetic_param_model(seed: int = 0, input_norm: str = "minmax", embed_dim: int = 512, fill=None):
    """Random-init AFx-Rep stand-in with the published architecture (cfg/model/pretext/
    param-panns-concat-l2.yaml:14-25) for benchmarks when the checkpoint is unavailable.
    `fill(model, seed)` may overwrite the weights (tests pass the oracle's deterministic fill)."""
    from .models.panns import Cnn14

    torch.manual_seed(seed)
    model = Cnn14(embed_dim, 48000, 2048, 1024, 128, 20, 20000, use_batchnorm=True, input_norm=input_norm)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                n = mod.num_features
                mod.running_mean.copy_(0.1 * torch.randn(n, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(n, generator=g))
                mod.weight.copy_(0.75 + 0.5 * torch.rand(n, generator=g))
                mod.bias.copy_(0.1 * torch.randn(n, generator=g))
            elif isinstance(mod, torch.nn.Conv2d):
                mod.weight.mul_(2.0 ** 0.5)  # keep activations O(1) through 12 ReLU layers
    if fill is not None:
        fill(model, seed)
    model.eval()
    if torch.cuda.is_available():
        model.cuda()
    model.reference_device = "cuda" if torch.cuda.is_available() else "cpu"  # as load_param_model(use_gpu=...) records it (get_param_embeds)
    return model


# ------------------- MIR feature extractor (reference utils.py:65-98) ---------------- #
def load_mir_feature_extractor(use_gpu: bool = False):
    class Model:
        def __init__(self) -> None:
            self.embed_dim = 49

    return Model()


def get_mir_feature_embeds(x: torch.Tensor, model, sample_rate: float, **kwargs):
    """Dictionary of hand-crafted features computed on the GPU (st_ito.features).  The reference
    (utils.py:76-98) calls compute_barkspectrum(x, sample_rate, mode="mono"), which binds the sample
    rate to `fft_size` (a 48 000-point FFT at the default 44.1 kHz filterbank); here the arguments
    are bound as its evaluation wrappers bind them (eval_pst.py:61-69): fft_size 32768 at `sample_rate`."""
    from . import features as F

    return {
        "lufs": F.compute_lufs(x, sample_rate),
        "rms": F.compute_rms_energy(x),
        "crest": F.compute_crest_factor(x),
        "barkspectrum": F.compute_barkspectrum(x, sample_rate=sample_rate, mode="mono"),
        "spectral_centroid": F.compute_spectral_centroid(x, sample_rate),
    }


# ------------------- audio feature (MFCC) extractor (reference utils.py:101-159) ---------------- #
class MFCCExtractor:
    """Device tables of torchaudio.transforms.MFCC(sample_rate=48000, n_mfcc=25, melkwargs={n_fft 2048,
    hop_length 1024, n_mels 128, center False}) -- restated from the library's published definition
    (un-vendored dependency, parity unpinned): Hann-window power STFT without padding, HTK mel filterbank
    (f_min 0, f_max sr/2, no normalisation), 10 log10(clamp(., 1e-10)) with an 80 dB floor below the item's
    maximum, orthonormal DCT-II.  The STFT + mel + log run in stito_logmel, the rest in stito_mfcc_stats."""

    def __init__(self, sample_rate: int = 48000, n_mfcc: int = 25, n_fft: int = 2048, hop_length: int = 1024, n_mels: int = 128):
        from .features import _create_triangular_filterbank_from

        _hip.require_gpu()
        self.sample_rate, self.n_mfcc, self.n_fft, self.hop, self.n_mels = sample_rate, n_mfcc, n_fft, hop_length, n_mels
        self.embed_dim = n_mfcc * 3
        self.device = dev = torch.device("cuda", torch.cuda.current_device())
        # HTK mel filterbank (n_freqs, n_mels): torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk")
        all_freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
        m_max = 2595.0 * math.log10(1.0 + (float(sample_rate // 2) / 700.0))
        m_pts = torch.linspace(0.0, m_max, n_mels + 2)
        f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
        fb = _create_triangular_filterbank_from(all_freqs, f_pts).numpy()
        m_start, m_len, m_off, m_w, m_stride = _hip.mel_tables(fb, dev)
        kk = np.arange(n_fft // 2)
        tw = np.stack([np.cos(-2 * np.pi * kk / n_fft), np.sin(-2 * np.pi * kk / n_fft)], 1).astype(np.float32)
        # orthonormal DCT-II (n_mels, n_mfcc): torchaudio.functional.create_dct(n_mfcc, n_mels, "ortho")
        n = torch.arange(float(n_mels))
        k = torch.arange(float(n_mfcc)).unsqueeze(1)
        dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
        self._keep = [torch.hann_window(n_fft, periodic=True).to(dev), torch.from_numpy(tw).to(dev).contiguous(), m_start, m_len,
                      m_off, m_w, dct.t().contiguous().to(dev)]
        FE = _hip.Frontend()
        FE.n_fft, FE.hop, FE.n_mels, FE.norm_mode, FE.no_center = n_fft, hop_length, n_mels, _hip.NORM_NONE, 1
        FE.window_dev, FE.twiddle_dev, FE.mel_start_dev, FE.mel_len_dev, FE.mel_off_dev, FE.mel_w_dev = (t.data_ptr() for t in self._keep[:6])
        FE.mel_w_stride = m_stride
        self.FE, self.dct = FE, self._keep[6]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        """(bs, chs, n) -> (bs, chs * 3 * n_mfcc): per channel [mean | std | max] over frames, rows L2-normalised."""
        L = _hip.lib()
        bs, chs, n = x.shape
        xin = x.detach().to(self.device, torch.float32).reshape(bs * chs, 1, n).contiguous()
        T = L.stito_num_frames_nocenter(n, self.n_fft, self.hop)
        if T < 2:
            raise ValueError("audio too short for MFCC statistics")
        lm = torch.empty((bs * chs, T, self.n_mels), dtype=torch.float32, device=self.device)
        _hip.check(L.stito_logmel(self.FE, _hip.ptr(xin), None, 0, bs * chs, 1, n, _hip.ptr(lm), _hip.stream_ptr()))
        out = torch.empty((bs, chs * 3 * self.n_mfcc), dtype=torch.float32, device=self.device)
        _hip.check(L.stito_mfcc_stats(_hip.ptr(lm), bs, chs, T, self.n_mels, _hip.ptr(self.dct), self.n_mfcc, 80.0, _hip.ptr(out),
                                      _hip.stream_ptr()))
        return out


def load_mfcc_feature_extractor(use_gpu: bool = False):
    """reference utils.py:101-113 (the extractor always lives on the GPU here)."""
    return MFCCExtractor()


def get_mfcc_feature_embeds(x: torch.Tensor, model, sample_rate: float, midside: bool = False, **kwargs):
    """reference utils.py:116-159: mono (channel mean) or mid/side (L+R, L-R) MFCC statistics, {"mono": (bs, E)}."""
    bs, chs, seq_len = x.shape
    if sample_rate != 48000:  # utils.py:130-131
        from .audio_io import resample

        x = resample(x, int(sample_rate), 48000)
    if chs == 2 and midside:
        x = torch.stack([x[:, 0, :] + x[:, 1, :], x[:, 0, :] - x[:, 1, :]], dim=1)
    else:
        x = x.mean(dim=1, keepdim=True)
    emb = model(x)
    return {"mono": emb.to(x.device).type_as(x)}

