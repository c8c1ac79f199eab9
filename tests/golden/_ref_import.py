"""Import pieces of the *reference* (/root/reference) inside the build container.

Used ONLY by tests/golden/make_golden.py to generate golden vectors.  The
reference's third-party deps (pedalboard, torchaudio, torchlibrosa, cma, ...)
are not installed here, so empty stand-in modules are registered in
sys.modules *just to let the import statements succeed*; no stand-in supplies
arithmetic that ends up in a fixture (anything that would need the real
package is simply not pinned -- see SURVEY.md section 8(c)).

Never imported by tests, bench.py or the product: /root/reference does not
exist on the GPU box.
"""
import importlib.machinery
import sys
import types

REF = "/root/reference"


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything

    def __call__(self, *a, **k):
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []  # behave like a package

    def _ga(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Anything

    m.__getattr__ = _ga  # type: ignore
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:  # make `import a.b.c as x` resolve through attribute access
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


def install_stubs():
    import torch

    for name in [
        "pedalboard", "torchaudio", "torchaudio.functional", "torchaudio.transforms",
        "torchaudio.compliance", "torchaudio.compliance.kaldi", "dasp_pytorch",
        "dasp_pytorch.functional", "pyloudnorm", "cma", "wav2clip", "laion_clap",
        "auraloss", "wandb", "timm", "timm.models", "timm.models.layers", "resampy",
        "soundfile", "numba", "umap", "transformers",
    ]:
        if name not in sys.modules:
            _stub(name)
    # pytorch_lightning: LightningModule must be a real class to subclass
    pl = _stub("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.LightningDataModule = object
    pl.Callback = object
    _stub("pytorch_lightning.callbacks")
    _stub("pytorch_lightning.cli")
    # torchlibrosa: front-end classes are supplied by the caller (see make_golden.py)
    tl = _stub("torchlibrosa")
    _stub("torchlibrosa.stft")
    _stub("torchlibrosa.augmentation")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return tl
