"""VST-like effect wrapper classes of the reference (st_ito/effects.py:783-985), as chain
descriptors for the HIP renderer.

Each Basic* class keeps the reference's duck-typed plugin protocol -- `.parameters`
(name -> Parameter with raw_value in [0,1], set_value, get_value) and
`.process(x, sample_rate)` on a (chs, n) float32 array -- but `.process` renders on the GPU
through libstito_hip (stito_render_population with a population of one).  Inside run_es the
instances are never called per candidate: the plugin dict is compiled once into a chain
descriptor (st_ito.engine.compile_chain) and the whole population is rendered at once.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import _hip


class Parameter:
    """reference st_ito/effects.py:784-797."""

    def __init__(self, init_value: float, min_value: float, max_value: float):
        self.min_value = min_value
        self.max_value = max_value
        self.set_value(init_value)

    def set_value(self, value: float):
        """Normalize the value to the range [0, 1] and store it as raw_value."""
        assert self.min_value <= value <= self.max_value
        self.raw_value = (value - self.min_value) / (self.max_value - self.min_value)

    def get_value(self):
        """Denormalize the value to the range [min_value, max_value]."""
        return self.raw_value * (self.max_value - self.min_value) + self.min_value


class _BasicEffect:
    KIND = -1
    NUM_CHANNELS = 1  # what run_optim.py:376-406 declares for this plugin

    def process(self, x: np.ndarray, sample_rate: float) -> np.ndarray:
        from .engine import render_single

        return render_single(self, np.asarray(x), sample_rate)


class BasicParametricEQ(_BasicEffect):
    """reference effects.py:800-873 -> parametric_eq (453-512) -> biqaud (395-450)."""

    KIND = _hip.FX_PARAMETRIC_EQ

    def __init__(
        self,
        low_shelf_gain_db: float = 0.0, low_shelf_cutoff_freq: float = 80.0, low_shelf_q_factor: float = 0.707,
        band0_gain_db: float = 0.0, band0_cutoff_freq: float = 300.0, band0_q_factor: float = 0.707,
        band1_gain_db: float = 0.0, band1_cutoff_freq: float = 1000.0, band1_q_factor: float = 0.707,
        band2_gain_db: float = 0.0, band2_cutoff_freq: float = 3000.0, band2_q_factor: float = 0.707,
        band3_gain_db: float = 0.0, band3_cutoff_freq: float = 10000.0, band3_q_factor: float = 0.707,
        high_shelf_gain_db: float = 0.0, high_shelf_cutoff_freq: float = 1000.0, high_shelf_q_factor: float = 0.707,
    ):
        self.parameters = OrderedDict([
            ("low_shelf_gain_db", Parameter(low_shelf_gain_db, -24.0, 24.0)),
            ("low_shelf_cutoff_freq", Parameter(low_shelf_cutoff_freq, 20.0, 4000.0)),
            ("low_shelf_q_factor", Parameter(low_shelf_q_factor, 0.1, 4.0)),
            ("band0_gain_db", Parameter(band0_gain_db, -24.0, 24.0)),
            ("band0_cutoff_freq", Parameter(band0_cutoff_freq, 20.0, 10000.0)),
            ("band0_q_factor", Parameter(band0_q_factor, 0.1, 4.0)),
            ("band1_gain_db", Parameter(band1_gain_db, -24.0, 24.0)),
            ("band1_cutoff_freq", Parameter(band1_cutoff_freq, 20.0, 10000.0)),
            ("band1_q_factor", Parameter(band1_q_factor, 0.1, 4.0)),
            ("band2_gain_db", Parameter(band2_gain_db, -24.0, 24.0)),
            ("band2_cutoff_freq", Parameter(band2_cutoff_freq, 20.0, 10000.0)),
            ("band2_q_factor", Parameter(band2_q_factor, 0.1, 4.0)),
            ("band3_gain_db", Parameter(band3_gain_db, -24.0, 24.0)),
            ("band3_cutoff_freq", Parameter(band3_cutoff_freq, 20.0, 10000.0)),
            ("band3_q_factor", Parameter(band3_q_factor, 0.1, 4.0)),
            ("high_shelf_gain_db", Parameter(high_shelf_gain_db, -24.0, 24.0)),
            ("high_shelf_cutoff_freq", Parameter(high_shelf_cutoff_freq, 200.0, 18000.0)),
            ("high_shelf_q_factor", Parameter(high_shelf_q_factor, 0.1, 4.0)),
        ])


class BasicCompressor(_BasicEffect):
    """reference effects.py:876-897 (pedalboard.Compressor = juce::dsp::Compressor<float>)."""

    KIND = _hip.FX_COMPRESSOR

    def __init__(self, threshold_db: float = 0.0, ratio: float = 4.0, attack_ms: float = 1.0,
                 release_ms: float = 100.0):
        self.parameters = OrderedDict([
            ("threshold_db", Parameter(threshold_db, -80.0, 0.0)),
            ("ratio", Parameter(ratio, 1.0, 20.0)),
            ("attack_ms", Parameter(attack_ms, 0.1, 100.0)),
            ("release_ms", Parameter(release_ms, 10.0, 1000.0)),
        ])


class BasicDistortion(_BasicEffect):
    """reference effects.py:900-916; like the reference the constructor ignores its arguments."""

    KIND = _hip.FX_DISTORTION

    def __init__(self, drive_db: float = 0.0, output_gain_db: float = 0.0):
        self.parameters = OrderedDict([
            ("drive_db", Parameter(0.0, -48.0, 48.0)),
            ("output_gain_db", Parameter(0.0, -24.0, 24.0)),
        ])


class BasicDelay(_BasicEffect):
    """reference effects.py:919-934 (pedalboard.Delay)."""

    KIND = _hip.FX_DELAY
    NUM_CHANNELS = 2

    def __init__(self, delay_seconds: float = 0.5, feedback: float = 0.5, mix: float = 0.5):
        self.parameters = OrderedDict([
            ("delay_seconds", Parameter(delay_seconds, 0.01, 1.0)),
            ("feedback", Parameter(feedback, 0.05, 1.0)),
            ("mix", Parameter(mix, 0.0, 1.0)),
        ])


class BasicReverb(_BasicEffect):
    """reference effects.py:937-959 (pedalboard.Reverb = juce::Reverb, Freeverb)."""

    KIND = _hip.FX_REVERB
    NUM_CHANNELS = 2

    def __init__(self, room_size: float = 0.5, damping: float = 0.5, wet_dry: float = 0.5, width: float = 0.5):
        self.parameters = OrderedDict([
            ("room_size", Parameter(room_size, 0.0, 1.0)),
            ("damping", Parameter(damping, 0.0, 1.0)),
            ("wet_dry", Parameter(wet_dry, 0.0, 1.0)),
            ("width", Parameter(width, 0.0, 1.0)),
        ])


class BasicGain(_BasicEffect):
    """The 'gain' stage of the benchmark chain EQ/comp/reverb/EQ/gain: x * 10^(gain_db/20) with
    gain_db in [-48, 48] as in the reference's apply_gain (effects.py:532-542)."""

    KIND = _hip.FX_GAIN

    def __init__(self, gain_db: float = 0.0):
        self.parameters = OrderedDict([("gain_db", Parameter(gain_db, -48.0, 48.0))])


def octave_band_filterbank(num_taps: int, sample_rate: float):
    """12 FIR filters (low-pass < 12 Hz... octave bands 31.25 Hz - 16 kHz ... high-pass > 18 kHz), the
    bank dasp_pytorch.noise_shaped_reverberation shapes its noise with: scipy.signal.firwin designs,
    time-reversed, float32, (12, 1, num_taps)."""
    import scipy.signal
    import torch

    filts = [scipy.signal.firwin(num_taps, 12, fs=sample_rate)]
    for fc in [31.25, 62.5, 125, 250, 500, 1000, 2000, 4000, 8000, 16000]:
        f_min, f_max = fc / np.sqrt(2), fc * np.sqrt(2)
        f_max = np.clip(f_max, a_min=0, a_max=(sample_rate / 2) * 0.999)
        filts.append(scipy.signal.firwin(num_taps, [f_min, f_max], fs=sample_rate, pass_zero=False))
    filts.append(scipy.signal.firwin(num_taps, 18000, fs=sample_rate, pass_zero=False))
    return torch.stack([torch.flip(torch.from_numpy(f.astype("float32")), dims=[0]) for f in filts], 0).unsqueeze(1)


def make_noise_bank(num_samples: int = 65536, num_bandpass_taps: int = 1023, sample_rate: float = 48000, seed: int = 0):
    """Band-filtered white noise (2, 12, num_samples) float32: what noise_shaped_reverberation draws
    afresh (unseeded) on every call, drawn once from a seeded generator so that the impulse
    responses -- and with them the optimisation -- are reproducible."""
    import torch

    g = torch.Generator().manual_seed(seed)
    wn = torch.randn(2, 12, num_samples + num_bandpass_taps - 1, generator=g)
    return torch.nn.functional.conv1d(wn, octave_band_filterbank(num_bandpass_taps, sample_rate), groups=12).contiguous()


class NoiseShapedReverb(_BasicEffect):
    """Convolution reverb of the reference's apply_reverb (effects.py:558-620 ->
    dasp_pytorch.noise_shaped_reverberation): 12 band gains, 12 band decays, mix -- all used raw in
    [0, 1] -- shape a noise impulse response of `num_samples` taps (65 536 in the library, 96 000
    = 2 s in BASELINE.json configs[4]) that is convolved with the audio.  Always stereo (mono is
    duplicated).  Not in run_optim.py's ES chains (the reference calls it from the autodiff path
    only); here it is a chain stage like any other."""

    KIND = _hip.FX_NOISE_REVERB
    NUM_CHANNELS = 2
    ALWAYS_STEREO = True

    def __init__(self, num_samples: int = 65536, num_bandpass_taps: int = 1023, sample_rate: float = 48000,
                 seed: int = 0, noise_bank=None):
        names = [f"band{b}_gain" for b in range(12)] + [f"band{b}_decay" for b in range(12)] + ["mix"]
        self.parameters = OrderedDict((n, Parameter(0.5, 0.0, 1.0)) for n in names)
        self.noise_bank = noise_bank if noise_bank is not None else make_noise_bank(num_samples, num_bandpass_taps,
                                                                                      sample_rate, seed)
        if self.noise_bank.dim() != 3 or tuple(self.noise_bank.shape[:2]) != (2, 12):
            raise ValueError("noise_bank must be (2, 12, num_samples)")
        self.num_samples = int(self.noise_bank.shape[-1])
        self._bank_dev = None

    def noise_bank_device(self, device):
        import torch

        if self._bank_dev is None or self._bank_dev.device != device:
            self._bank_dev = self.noise_bank.to(device, torch.float32).contiguous()
        return self._bank_dev


class BasicChorus(_BasicEffect):
    """reference effects.py:962-985 (pedalboard.Chorus = juce::dsp::Chorus<float>).  `rate_hz` is a declared parameter that
    the reference's process() does not pass on: the LFO runs at the library default, 1 Hz."""

    KIND = _hip.FX_CHORUS

    def __init__(self, rate_hz: float = 1.0, centre_delay_ms: float = 7.0, depth: float = 0.1, feedback: float = 0.5,
                 mix: float = 0.5):
        self.parameters = OrderedDict([
            ("rate_hz", Parameter(rate_hz, 0.1, 10.0)),
            ("centre_delay_ms", Parameter(centre_delay_ms, 0.1, 20.0)),
            ("depth", Parameter(depth, 0.0, 1.0)),
            ("feedback", Parameter(feedback, 0.0, 1.0)),
            ("mix", Parameter(mix, 0.0, 1.0)),
        ])


_CHORUS_LFO = {}


def chorus_lfo_device(sample_rate: float, n_samples: int, device):
    """LFO table of the chorus stage (stito_chorus_lfo: juce's float phase recurrence, one serial walk on the GPU), cached per
    (sample rate, device) and grown in powers of two."""
    import torch

    rate_hz = 1.0  # the reference's BasicChorus.process() never passes rate_hz on (effects.py:962-985): the library default
    key = (float(sample_rate), rate_hz, str(device))
    t = _CHORUS_LFO.get(key)
    if t is None or t.numel() < n_samples:
        n = 1 << max(16, int(n_samples - 1).bit_length())
        t = torch.empty(n, dtype=torch.float32, device=device)
        _hip.check(_hip.lib().stito_chorus_lfo(float(sample_rate), rate_hz, n, _hip.ptr(t), _hip.stream_ptr()))
        _CHORUS_LFO[key] = t
    return t


BASIC_CHAINS = {
    # scripts/run_optim.py:375-407 (--effect-type basic)
    "basic": [("ParametricEQ", BasicParametricEQ, 1), ("Compressor", BasicCompressor, 1),
              ("Distortion", BasicDistortion, 1), ("Delay", BasicDelay, 2), ("Reverb", BasicReverb, 2)],
    # BASELINE.json configs[0]: 2-effect chain
    "eq-comp": [("ParametricEQ", BasicParametricEQ, 1), ("Compressor", BasicCompressor, 1)],
    # BASELINE.json configs[1]: 5-effect chain EQ/comp/reverb/EQ/gain (D = 45)
    "bench5": [("ParametricEQ", BasicParametricEQ, 1), ("Compressor", BasicCompressor, 1),
               ("Reverb", BasicReverb, 2), ("ParametricEQ2", BasicParametricEQ, 1), ("Gain", BasicGain, 1)],
    "eq": [("ParametricEQ", BasicParametricEQ, 1)],
    # BASELINE.json configs[4] shape: convolution reverb in the chain (class_path may be a
    # functools.partial(NoiseShapedReverb, num_samples=96000) for the 2 s impulse response)
    "eq-convreverb-gain": [("ParametricEQ", BasicParametricEQ, 1), ("ConvReverb", NoiseShapedReverb, 2),
                           ("Gain", BasicGain, 1)],
}


def make_plugins(chain="basic", with_bypass: bool = False):
    """Build a plugin dict in the reference's schema.  with_bypass=False follows
    run_optim.py:409-437; with_bypass=True follows load_plugins (style_transfer.py:17-42)."""
    spec = BASIC_CHAINS[chain] if isinstance(chain, str) else chain
    plugins = OrderedDict()
    for name, cls, nch in spec:
        inst = cls()
        names = list(inst.parameters.keys())
        if with_bypass:
            names = ["our_bypass"] + names
        plugins[name] = {"class_path": cls, "num_params": len(names), "num_channels": nch,
                         "fixed_parameters": {}, "instance": inst, "parameter_names": names}
    return plugins
