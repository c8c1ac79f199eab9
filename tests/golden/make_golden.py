"""Generate golden vectors by running the REFERENCE's own code (/root/reference).

Run once in the build container (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs small .npz files next to this script; they are data only (inputs and
expected outputs).  Third-party packages the reference imports but that are
absent here are replaced by empty stand-ins (tests/golden/_ref_import.py); the
only stand-in that contributes arithmetic is the torchlibrosa front-end, for which
this repo's own restatement (oracle/st_ito_oracle.py: Spectrogram, LogmelFilterBank)
is supplied -- so G6/G7 pin the reference's conv/BN/pool/FC/loss code and its
orchestration, not the front-end ("parity unpinned" there).

  G1 eq_biquad.npz        effects.py:395-450  biqaud coefficient grid
  G2 eq_parametric.npz    effects.py:453-512  parametric_eq on noise + impulse
  G3 process_audio.npz    style_transfer.py:17-115  channel rules / bypass / peak norm
  G4 params_to_dict.npz   style_transfer.py:324-359
  G5 param_embeds_toy.npz utils.py:444-508 with a deterministic toy model
  G6 cnn14_trunk_*.npz    panns.py:209-281 conv stack (+ this repo's front-end)
  G6b cnn14_trunk_minmax_262144.npz  the same on a 257 x 128 map (input stored as its synth_audio recipe)
  G7 evaluate_*.npz       style_transfer.py:399-692 run_es -> evaluate losses (fake `cma`)
  G9 savepop_reference.npz  style_transfer.py:362-396 savepop_to_disk with the embedding dict run_es hands it (two files)
  G8 features.npz         features.py:166-264 bark spectrum (3 modes, 2 FFT sizes), RMS, crest factor
                          on O.synth_audio(seed, 2, n) inputs (the fixture stores the recipe, not the audio)
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import _ref_import  # noqa: E402

_ref_import.install_stubs()
import st_ito_oracle as O  # noqa: E402

# supply this repo's front-end restatement where the reference imports torchlibrosa
sys.modules["torchlibrosa.stft"].Spectrogram = O.Spectrogram
sys.modules["torchlibrosa.stft"].LogmelFilterBank = O.LogmelFilterBank
sys.modules["torchlibrosa.augmentation"].SpecAugmentation = lambda **k: torch.nn.Identity()

import st_ito.effects as RE  # noqa: E402  (reference)
import st_ito.style_transfer as RS  # noqa: E402
import st_ito.utils as RU  # noqa: E402
import st_ito.models.panns as RP  # noqa: E402

SR = 48000


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrs.items()),
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def g1():
    rows, outs = [], []
    for kind_i, kind in enumerate(["low_shelf", "peaking", "high_shelf"]):
        for g in [-24.0, -6.0, 0.0, 3.0, 24.0]:
            for f in [20.0, 80.0, 1000.0, 10000.0, 18000.0]:
                for q in [0.1, 0.707, 4.0]:
                    b, a = RE.biqaud(g, f, q, SR, kind)
                    rows.append([kind_i, g, f, q])
                    outs.append(np.concatenate([b, a]))
    save("eq_biquad.npz", args=np.array(rows), ba=np.array(outs), sample_rate=np.array(SR))


def _eq_param_sets(rng):
    eq = RE.BasicParametricEQ()
    names = list(eq.parameters.keys())
    lo = np.array([eq.parameters[n].min_value for n in names])
    hi = np.array([eq.parameters[n].max_value for n in names])
    sets = [np.array([eq.parameters[n].get_value() for n in names])]  # defaults (identity)
    ext = lo.copy()
    ext[0::3] = 24.0; ext[1::3] = lo[1::3]; ext[2::3] = 4.0            # 20 Hz / Q 4 / +24 dB
    sets.append(ext)
    ext2 = hi.copy(); ext2[0::3] = -24.0; ext2[2::3] = 0.1
    sets.append(ext2)
    ext3 = lo.copy(); ext3[0::3] = -24.0; ext3[2::3] = 4.0
    sets.append(ext3)
    for _ in range(6):
        sets.append(lo + rng.random(len(names)) * (hi - lo))
    return names, np.array(sets)


def g2():
    rng = np.random.default_rng(11)
    names, sets = _eq_param_sets(rng)
    n = 8192
    noise = (0.25 * rng.standard_normal((1, n))).astype(np.float32)
    imp = np.zeros((1, n), np.float32); imp[0, 0] = 1.0
    ys_noise, ys_imp = [], []
    for p in sets:
        kw = dict(
            low_shelf_gain_db=p[0], low_shelf_cutoff_freq=p[1], low_shelf_q_factor=p[2],
            band_gains_db=[p[3], p[6], p[9], p[12]], band_cutoff_freqs=[p[4], p[7], p[10], p[13]],
            band_q_factors=[p[5], p[8], p[11], p[14]],
            high_shelf_gain_db=p[15], high_shelf_cutoff_freq=p[16], high_shelf_q_factor=p[17],
        )
        ys_noise.append(RE.parametric_eq(noise, SR, **kw))
        ys_imp.append(RE.parametric_eq(imp, SR, **kw))
    save("eq_parametric.npz", params=sets, noise=noise, y_noise=np.array(ys_noise),
         y_impulse=np.array(ys_imp), sample_rate=np.array(SR))


def _ref_plugins(kinds, with_bypass):
    plugins = OrderedDict()
    for i, k in enumerate(kinds):
        name = "ParametricEQ" if i == 0 else f"ParametricEQ{i + 1}"
        plugins[name] = {"class_path": RE.BasicParametricEQ, "num_params": None,
                         "num_channels": 1, "fixed_parameters": {}}
    if with_bypass:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            plugins, total, init = RS.load_plugins(plugins)
    else:  # run_optim.py:409-437
        total = 0
        for name, plugin in plugins.items():
            inst = plugin["class_path"]()
            plugin["num_params"] = len(inst.parameters)
            plugin["instance"] = inst
            plugin["parameter_names"] = list(inst.parameters.keys())
            total += plugin["num_params"]
    return plugins, total


def g3_g4():
    rng = np.random.default_rng(5)
    n = 6000
    out = {}
    cases = []
    for ci, (nplug, bypass, chs) in enumerate([(1, False, 1), (1, True, 2), (2, False, 2), (2, True, 1)]):
        plugins, total = _ref_plugins(["eq"] * nplug, bypass)
        x = (0.3 * rng.standard_normal((chs, n))).astype(np.float32)
        w = rng.random(total)
        if bypass:
            w[0] = 0.9  # > 0.5: would "bypass" if the flag were honoured (it is not)
        y = RS.process_audio(x.copy(), w, SR, plugins)
        d = RS.parameters_to_dict(w, plugins)
        flat = np.array([v for pn in d for v in d[pn].values()], dtype=np.float64)
        out[f"x{ci}"], out[f"w{ci}"], out[f"y{ci}"], out[f"d{ci}"] = x, w, y, flat
        cases.append([nplug, int(bypass), chs])
    out["cases"] = np.array(cases)
    # fixed_parameters path (style_transfer.py:79-84)
    plugins, total = _ref_plugins(["eq"], False)
    plugins["ParametricEQ"]["fixed_parameters"] = {"band1_gain_db": 12.0, "band1_cutoff_freq": 2500.0}
    x = (0.3 * rng.standard_normal((1, n))).astype(np.float32)
    w = rng.random(total)
    out["xf"], out["wf"] = x, w
    out["yf"] = RS.process_audio(x.copy(), w, SR, plugins)
    save("process_audio.npz", **out)


class ToyModel(torch.nn.Module):
    """Deterministic (mid, side) 'model' to exercise utils.py:444-508 post-processing."""

    def __init__(self, nan_mode=0):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.nan_mode = nan_mode

    def forward(self, x):
        mid = torch.stack([x[:, 0, :8] * 3 + 1, x[:, -1, 8:16] - 2], 1).flatten(1)
        side = torch.stack([x[:, 0, 16:24], x[:, -1, 24:32] * 5], 1).flatten(1)
        if self.nan_mode == 1:
            mid = mid.clone(); mid[0, 0] = float("nan")
        if self.nan_mode == 2:
            side = side.clone(); side[0, 1] = float("nan")
        return mid, side


def g5():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((3, 2, 64)) * np.array([0.1, 2.0, 1e-10])[:, None, None]).astype(np.float32)
    out = {"x": x}
    for mode in (0, 1, 2):
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            e = RU.get_param_embeds(torch.from_numpy(x.copy()), ToyModel(mode), SR)
        out[f"mid{mode}"], out[f"side{mode}"] = e["mid"].numpy(), e["side"].numpy()
    save("param_embeds_toy.npz", **out)


def _ref_cnn14(input_norm, seed=0):
    m = RP.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, use_batchnorm=True, input_norm=input_norm)
    O.fill_deterministic(m, seed)
    m.eval()
    return m


def g6():
    n = 32768
    x = torch.stack([O.synth_audio(21, 2, n), O.synth_audio(22, 2, n)], 0)  # (2,2,n)
    xm = O.synth_audio(23, 1, n)[None]                                       # (1,1,n) mono
    for norm in ("minmax", "batchnorm", "none"):
        m = _ref_cnn14(norm)
        with torch.no_grad():
            mid, side = m(x)
            midm, sidem = m(xm)
            e = RU.get_param_embeds(x.clone(), m, SR)
            om = O.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, norm)
            om.load_state_dict(m.state_dict())
            om.eval()
            lm = om.logmel(x)
        save(f"cnn14_trunk_{norm}.npz", x=x.numpy(), x_mono=xm.numpy(), logmel=lm.numpy(),
             mid=mid.numpy(), side=side.numpy(), mid_mono=midm.numpy(), side_mono=sidem.numpy(),
             embed_mid=e["mid"].numpy(), embed_side=e["side"].numpy(), seed=np.array(0))


class FakeES:
    """Stand-in for cma.CMAEvolutionStrategy: replays a fixed population, records fvals."""
    W = None
    told = None

    def __init__(self, w0, sigma0, opts):
        self.result = (None, float("inf"))

    def ask(self):
        return [w.copy() for w in FakeES.W]

    def tell(self, W, fvals):
        FakeES.told = list(fvals)
        i = int(np.argmin(fvals))
        self.result = (W[i], fvals[i])

    def disp(self):
        pass


def g7():
    sys.modules["cma"].CMAEvolutionStrategy = FakeES
    rng = np.random.default_rng(17)
    m = _ref_cnn14("minmax")
    for tag, chs, n, P in (("stereo", 2, 40000, 3), ("mono", 1, 36000, 2)):
        plugins, total = _ref_plugins(["eq"], False)
        x = O.synth_audio(31, chs, n)[None]
        tgt = O.synth_audio(32, chs, n)[None]
        FakeES.W = [rng.random(total) for _ in range(P)]
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            res = RS.run_es(x.clone(), tgt.clone(), SR, plugins, m, RU.get_param_embeds,
                            max_iters=1, popsize=P, find_w0=False, sigma0=0.33)
        save(f"evaluate_{tag}.npz", x=x.numpy(), target=tgt.numpy(), W=np.array(FakeES.W),
             fvals=np.array(FakeES.told, dtype=np.float64), wopt=np.asarray(res["wopt"]),
             fopt=np.array(res["fopt"]), output_audio=res["output_audio"].numpy(), seed=np.array(0))

def g6b():
    """One bench-shaped map through the reference's own conv stack: n = 262 144 -> 257 x 128 log-mel,
    final map 8 x 4.  minmax only; the fixture stores the input's recipe (O.synth_audio(seed, 2, n)), not the audio."""
    n, seed = 262144, 24
    x = O.synth_audio(seed, 2, n)[None]
    m = _ref_cnn14("minmax")
    with torch.no_grad():
        mid, side = m(x)
        e = RU.get_param_embeds(x.clone(), m, SR)
    save("cnn14_trunk_minmax_262144.npz", audio_seed=np.array(seed), n=np.array(n), mid=mid.numpy(), side=side.numpy(),
         embed_mid=e["mid"].numpy(), embed_side=e["side"].numpy(), seed=np.array(0))


def g8():
    import st_ito.features as RF  # reference (torchaudio / pyloudnorm are stand-ins: centroid and LUFS are not pinned)
    seeds, n = [201, 202, 203], 70000
    x = torch.stack([O.synth_audio(sd, 2, n) * (0.3 + 0.3 * i) for i, sd in enumerate(seeds)])
    out = dict(seeds=np.array(seeds), n=np.array(n), scales=np.array([0.3 + 0.3 * i for i in range(3)]))
    for fft in (32768, 4096):
        for mode in ("mono", "stereo", "mid-side"):
            out[f"bark_{fft}_{mode.replace('-', '')}"] = RF.compute_barkspectrum(x, fft_size=fft, sample_rate=SR, mode=mode).numpy()
    out["bark_fb_32768"] = RF.barkscale_fbanks(32768 // 2 + 1, 20.0, 20000.0, 24, SR).numpy().astype(np.float32)[::64]  # every 64th row
    out["rms"] = RF.compute_rms_energy(x).numpy()
    out["crest"] = RF.compute_crest_factor(x).numpy()
    save("features.npz", **out)


def g9():
    """The reference's savepop_to_disk (style_transfer.py:362-396) as it IS: run_es hands it the embedding DICT, the zip over
    (fvals, audios, dict) stops at the dict's two keys, so two files per population are written -- candidates 0 and 1, ordered by
    their own fitness.  torchaudio.save is a recorder here (file name, tensor as handed over, rate)."""
    import tempfile
    rng = np.random.default_rng(9)
    P, chs, n = 5, 2, 257
    fvals = [float(v) for v in -rng.random(P)]
    fvals[0] = max(fvals) + 0.125          # candidate 0 is the worst of the five, candidate 1 somewhere in the middle
    audios = [torch.from_numpy(rng.standard_normal((1, chs, n)).astype(np.float32) * (0.2 + 0.3 * i)) for i in range(P)]
    embeds = {"mid": torch.zeros(P, 4), "side": torch.zeros(P, 4)}
    rec = []
    sys.modules["torchaudio"].save = lambda path, t, sr, backend=None: rec.append((os.path.basename(path), t.clone().numpy(), sr, backend))
    with tempfile.TemporaryDirectory() as d:
        RS.savepop_to_disk(3, list(fvals), embeds, [a.clone() for a in audios], d, SR)
        made = sorted(os.listdir(d))
    assert made == ["pop_3"] and len(rec) == 2, (made, len(rec))
    save("savepop_reference.npz", fvals=np.array(fvals), audios=np.stack([a.numpy() for a in audios]),
         names=np.array([r[0] for r in rec]), written=np.stack([r[1] for r in rec]), rate=np.array(rec[0][2]),
         backend=np.array(str(rec[0][3])), iteration=np.array(3))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g5", "g6", "g6b", "g7", "g8", "g9"]
    for name in which:
        {"g1": g1, "g2": g2, "g3": g3_g4, "g5": g5, "g6": g6, "g6b": g6b, "g7": g7, "g8": g8, "g9": g9}[name]()
