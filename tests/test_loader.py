"""load_param_model (reference st_ito/utils.py:511-551): Lightning-style checkpoint + config.yaml round trip.

The published afx-rep.ckpt is not available offline, so the checkpoint is synthesised in the layout the
reference reads: `state_dict` with the encoder under the `encoder.` prefix (first occurrence stripped,
utils.py:539-542) next to foreign keys that must be skipped, and a config.yaml whose
model.init_args.encoder.class_path is `lcap.models.panns.Cnn14` (cfg/model/pretext/param-panns-concat-l2.yaml:14-25;
`lcap` -> `st_ito`, utils.py:531)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import yaml

import st_ito_oracle as O

SR = 48000


class _Booby:
    """A foreign class inside the checkpoint (Lightning hyper-parameters): loading must never run its code."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __setstate__(self, state):
        raise AssertionError("checkpoint loading executed code of a pickled foreign class")


def _write_checkpoint(tmp_path, input_norm, use_batchnorm=True, seed=3, tamper=None):
    om = O.make_synthetic_model(seed, input_norm=input_norm, use_batchnorm=use_batchnorm)
    sd = {f"encoder.{k}": v.clone() for k, v in om.state_dict().items()}
    if tamper:
        tamper(sd)
    # foreign keys the loader must skip: projector / classifier heads of the pre-training system (methods/param.py)
    sd["projector.0.weight"] = torch.randn(8, 512)
    sd["inst_classifier.weight"] = torch.randn(63, 512)
    mod = types.ModuleType("fake_lightning_pkg")
    _Booby.__module__ = "fake_lightning_pkg"
    mod._Booby = _Booby
    sys.modules["fake_lightning_pkg"] = mod
    ckpt = {"epoch": 12, "global_step": 3456, "pytorch-lightning_version": "2.1.0", "state_dict": sd,
            "hyper_parameters": _Booby(lr=1e-4, num_instances=63), "optimizer_states": [{"state": {}, "param_groups": []}],
            # objects Lightning checkpoints commonly carry and the reference's torch.load reads: they are pickled
            # through REDUCE with arguments, so their stand-ins must accept (and drop) constructor arguments
            "hparams_extra": {"ckpt_dir": __import__("pathlib").PosixPath("/tmp/run/ckpts"), "lr_np": np.float64(1e-4),
                              "steps_np": np.int64(7)}}
    d = tmp_path / f"ckpt_{input_norm}_{use_batchnorm}"
    d.mkdir()
    torch.save(ckpt, str(d / "afx-rep.ckpt"))
    cfg = {"model": {"class_path": "lcap.methods.param.ParameterEstimator",
                     "init_args": {"lr": 1e-4, "num_instances": 63, "embed_mode": "concat", "norm": "L2",
                                   "encoder": {"class_path": "lcap.models.panns.Cnn14",
                                               "init_args": {"embed_dim": 512, "sample_rate": 48000, "window_size": 2048,
                                                             "hop_size": 1024, "mel_bins": 128, "fmin": 20, "fmax": 20000,
                                                             "use_batchnorm": use_batchnorm, "input_norm": input_norm}}}}}
    with open(d / "config.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    return str(d / "afx-rep.ckpt"), om


@pytest.mark.parametrize("input_norm", ["minmax", "batchnorm", "none"])
def test_load_param_model_strict_roundtrip_host(tmp_path, input_norm):
    from st_ito.models.panns import Cnn14
    from st_ito.utils import load_param_model
    path, om = _write_checkpoint(tmp_path, input_norm)
    model = load_param_model(path, use_gpu=False)
    assert isinstance(model, Cnn14) and not model.training and model.input_norm == input_norm
    got = model.state_dict()
    ref = om.state_dict()
    assert list(got.keys()) == list(ref.keys())          # strict: same keys, foreign ones skipped
    for k in ref:
        assert torch.equal(got[k].cpu(), ref[k]), k
    # the torchlibrosa front-end parameters travel in the checkpoint (frozen Parameters there)
    for k in ("spectrogram_extractor.stft.conv_real.weight", "spectrogram_extractor.stft.conv_imag.weight",
              "logmel_extractor.melW", "bn0.running_mean", "conv_block6.bn2.num_batches_tracked", "fc_side.bias"):
        assert k in got


def test_load_param_model_errors(tmp_path):
    from st_ito.utils import load_param_model
    with pytest.raises(FileNotFoundError):               # no network here: the reference's wget is not attempted
        load_param_model(str(tmp_path / "missing" / "afx-rep.ckpt"))
    path, _ = _write_checkpoint(tmp_path, "minmax", tamper=lambda sd: sd.pop("encoder.fc_mid.bias"))
    with pytest.raises(RuntimeError):                    # strict load_state_dict like the reference (utils.py:544)
        load_param_model(path)
    # use_batchnorm=False checkpoints have no bn1/bn2 tensors (panns.py:52-58)
    path2, om2 = _write_checkpoint(tmp_path, "none", use_batchnorm=False)
    m2 = load_param_model(path2)
    assert not any(".bn1." in k or ".bn2." in k for k in m2.state_dict())


@pytest.mark.gpu
@pytest.mark.parametrize("input_norm", ["minmax", "batchnorm", "none"])
def test_load_param_model_gpu_embeddings(tmp_path, input_norm):
    """The loaded model's embeddings are bitwise those of a Cnn14 built directly from the same weights, and within
    1e-4 of the oracle's torch-CPU forward of the source module."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito.models.panns import Cnn14
    from st_ito.utils import get_param_embeds, load_param_model
    path, om = _write_checkpoint(tmp_path, input_norm)
    model = load_param_model(path, use_gpu=True)
    assert next(model.parameters()).is_cuda
    direct = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, input_norm)
    direct.load_state_dict(om.state_dict())
    direct.eval().cuda()
    x = torch.stack([O.synth_audio(5, 2, 70000), O.synth_audio(6, 2, 70000) * 0.3])
    e1 = get_param_embeds(x.clone(), model, SR)
    e2 = get_param_embeds(x.clone(), direct, SR)
    e_ref = O.get_param_embeds(x.clone(), om, SR)
    for k in ("mid", "side"):
        assert e1[k].shape == (2, 512) and not e1[k].is_cuda      # returned on the input's device (utils.py:503-506)
        assert torch.equal(e1[k], e2[k])
        assert (e1[k] - e_ref[k]).abs().max() / e_ref[k].abs().max() < 1e-4


@pytest.mark.gpu
def test_front_end_that_is_not_a_windowed_dft_is_rejected(tmp_path):
    """The HIP front end computes the STFT by FFT; a checkpoint whose conv_real/conv_imag kernels are not
    window x DFT matrix cannot be honoured and must fail loudly (models/panns.py prepare())."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito.utils import get_param_embeds, load_param_model

    def tamper(sd):
        sd["encoder.spectrogram_extractor.stft.conv_imag.weight"][7] += 0.01
    path, _ = _write_checkpoint(tmp_path, "minmax", tamper=tamper)
    model = load_param_model(path, use_gpu=True)
    with pytest.raises(NotImplementedError):
        get_param_embeds(O.synth_audio(5, 2, 70000)[None], model, SR)
    # a different (but valid) analysis window is honoured: Hamming instead of Hann
    def hamming(sd):
        n = 2048
        w_new = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n) / n)
        for k in ("conv_real", "conv_imag"):
            key = f"encoder.spectrogram_extractor.stft.{k}.weight"
            nb = sd[key].shape[0]
            kk = np.arange(nb)[:, None] * np.arange(n)[None, :]
            ang = -2 * np.pi * (kk % n) / n
            base = np.cos(ang) if k == "conv_real" else np.sin(ang)
            sd[key] = torch.from_numpy((base * w_new[None, :]).astype(np.float32))[:, None, :]
    path2, om2 = _write_checkpoint(tmp_path, "none", tamper=hamming, seed=4)
    m2 = load_param_model(path2, use_gpu=True)
    sd = {f"encoder.{k}": v.clone() for k, v in om2.state_dict().items()}
    hamming(sd)
    om2.load_state_dict({k[len("encoder."):]: v for k, v in sd.items()})   # the oracle convolves with the stored kernels
    x = O.synth_audio(5, 2, 70000)[None]
    e, e_ref = get_param_embeds(x.clone(), m2, SR), O.get_param_embeds(x.clone(), om2, SR)
    for k in ("mid", "side"):
        assert (e[k] - e_ref[k]).abs().max() / e_ref[k].abs().max() < 1e-4
