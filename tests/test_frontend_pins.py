"""Pins for the oracle's restatement of the torchlibrosa / librosa front end (reference st_ito/models/panns.py:147-168, 230-231).

torchlibrosa and librosa are not in the image, so the oracle restates them from their published definitions; two INDEPENDENT
implementations of the same definitions are installed here -- torch.stft and transformers.audio_utils.mel_filter_bank (the
librosa-compatible Slaney filter bank of the transformers library) -- and these tests hold the restatement to them.  The MFCC /
bark metric paths use the same mel filter bank constructor and torch.stft-equivalent transforms (tests/test_oracle_golden.py
pins those against the reference's own features.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import st_ito_oracle as O  # noqa: E402


@pytest.mark.parametrize("n_fft,hop,n", [(2048, 1024, 48000), (2048, 1024, 30001), (1024, 512, 20000), (512, 128, 9000)])
def test_spectrogram_matches_torch_stft(n_fft, hop, n):
    """Spectrogram(power=2, center=True, pad_mode='reflect', periodic Hann) as a windowed-DFT conv1d == |torch.stft|^2."""
    x = torch.stack([O.synth_audio(3, 1, n)[0], 0.3 * O.synth_audio(4, 1, n)[0]])
    with torch.no_grad():
        got = O.Spectrogram(n_fft, hop)(x)[:, 0].double()                      # (N, T, F)
    ref = torch.stft(x.double(), n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64),
                     center=True, pad_mode="reflect", return_complex=True).abs().pow(2).transpose(1, 2)
    assert got.shape == ref.shape == (2, n // hop + 1, n_fft // 2 + 1)
    # the restatement multiplies by a float32 DFT matrix (as torchlibrosa does); relative to each frame's strongest bin
    scale = ref.amax(dim=2, keepdim=True).clamp_min(1e-30)
    assert ((got - ref).abs() / scale).max().item() < 1e-5


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(48000, 2048, 128, 20, 20000), (48000, 1024, 64, 20, 20000),
                                                       (44100, 2048, 128, 0, 22050), (16000, 512, 40, 50, 8000)])
def test_mel_filterbank_matches_transformers(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm='slaney') == transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')."""
    tau = pytest.importorskip("transformers.audio_utils")
    ref = tau.mel_filter_bank(n_fft // 2 + 1, n_mels, fmin, fmax, sr, norm="slaney", mel_scale="slaney")   # (n_bins, n_mels) float64
    got = O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax).T.astype(np.float64)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 1e-7   # the restatement rounds to float32 once (librosa's dtype)


def test_logmel_matches_independent_chain():
    """The whole log-mel chain of panns.py:230-231 (power spectrogram -> mel -> 10 log10 clamp 1e-10) from the two independent pieces."""
    tau = pytest.importorskip("transformers.audio_utils")
    n = 48000
    x = O.synth_audio(5, 1, n)
    with torch.no_grad():
        got = O.LogmelFilterBank(48000, 2048, 128, 20, 20000)(O.Spectrogram(2048, 1024)(x))[0, 0].double()   # (T, 128)
    pw = torch.stft(x.double(), 2048, hop_length=1024, window=torch.hann_window(2048, periodic=True, dtype=torch.float64), center=True,
                    pad_mode="reflect", return_complex=True).abs().pow(2)[0].T
    mel = pw @ torch.from_numpy(tau.mel_filter_bank(1025, 128, 20, 20000, 48000, norm="slaney", mel_scale="slaney"))
    ref = 10.0 * torch.log10(mel.clamp_min(1e-10))
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 2e-3   # dB; float32 matrix products against float64
