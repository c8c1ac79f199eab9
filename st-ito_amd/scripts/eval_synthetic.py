#!/usr/bin/env python
"""Synthetic style-transfer benchmark on the MI355X path -- the ES arm of the reference's scripts/eval/eval_synthetic.py (methods
table 158-206, loop 263-456): every dry example is paired with a randomly chosen OTHER example of the same source type from each
test case (the same chain applied to different material), run_es carries the style over with the harness's settings (popsize 128,
32 iterations, sigma0 0.33, find_w0 False, random_crop False), and the result is scored against the dry example's own rendering
in that test case (the ground truth): multi-resolution STFT error, the same after peak normalisation, and the style metric
(mean cosine similarity of the AFx-Rep embeddings) against ground truth and against the target.  Output, target and ground truth
are cropped to their common length, brought to -22 LUFS and saved; results.json is rewritten after every example.

Differences, all stated: only the `style-es (param-panns)` method on the `pb` plugin set and the `input` reference row are built
(random / rule-based / DeepAFx-ST are other methods, the `vst` set needs binary plugins); auraloss is an un-vendored dependency
absent here, its MultiResolutionSTFTLoss() defaults are restated (`mrstft_error`; unpinned); `--seed` seeds the draws and the
CMA-ES.  Directory layout as the reference reads it: <input_dir>/dry/*.wav and <input_dir>/<test_case>/*.wav with the same file
names, the source type ("music" / "vocals" / "straight" / "speech") in the name.

    python st-ito_amd/scripts/eval_synthetic.py <input_dir> --output_dir out --max-iters 32 --popsize 128
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

TEST_CASES = ["easy-1", "easy-2", "medium-1", "medium-2", "hard-1", "hard-2"]   # eval_synthetic.py:73-80


def get_source_type(filename: str) -> str:
    """eval_synthetic.py:45-54."""
    if "music" in filename:
        return "music"
    elif "vocals" in filename or "straight" in filename:
        return "vocals"
    elif "speech" in filename:
        return "speech"
    raise ValueError(f"Unknown source type for {filename}")


def get_pb_plugins():
    """The `pb_plugins` table of eval_synthetic.py:103-134: EQ, compressor, distortion (1 channel), delay, reverb (2 channels)."""
    from st_ito.effects import BasicCompressor, BasicDelay, BasicDistortion, BasicParametricEQ, BasicReverb

    spec = (("ParametricEQ", BasicParametricEQ, 1), ("Compressor", BasicCompressor, 1), ("Distortion", BasicDistortion, 1),
            ("Delay", BasicDelay, 2), ("Reverb", BasicReverb, 2))
    return OrderedDict((n, {"class_path": c, "num_params": None, "num_channels": ch, "fixed_parameters": {}}) for n, c, ch in spec)


def mrstft_error(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """auraloss.freq.MultiResolutionSTFTLoss()(x, y) with the library's defaults, restated (auraloss is absent: unpinned): FFT
    sizes 1024 / 2048 / 512 with hops 120 / 240 / 50 and Hann windows of 600 / 1200 / 240; per resolution the spectral
    convergence ||Y| - |X||_F / ||Y||_F (per (item, channel) row, then the mean over rows: auraloss 0.4.0) plus the mean absolute difference of the log magnitudes (magnitudes clamped at
    sqrt(1e-8)); the mean over the three.  x, y: (bs, chs, n) of equal shape; x is the estimate, y the reference."""
    assert x.shape == y.shape and x.dim() == 3
    xs, ys = x.reshape(-1, x.shape[-1]).to(torch.float32), y.reshape(-1, y.shape[-1]).to(torch.float32)
    total = 0.0
    for n_fft, hop, win in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
        w = torch.hann_window(win, device=xs.device)
        mags = []
        for s in (xs, ys):
            z = torch.stft(s, n_fft, hop, win, w, return_complex=True)
            mags.append(torch.sqrt(torch.clamp(z.real ** 2 + z.imag ** 2, min=1e-8)))
        xm, ym = mags
        # auraloss >= 0.3 (SpectralConvergenceLoss): the Frobenius norms are taken PER ROW over (bins, frames), then averaged --
        # for stereo pairs of unequal channel energy that is not one norm over the flattened batch
        sc = (torch.linalg.norm(ym - xm, dim=(-2, -1)) / torch.linalg.norm(ym, dim=(-2, -1))).mean()
        lm = torch.mean(torch.abs(torch.log(xm) - torch.log(ym)))
        total = total + sc + lm
    return total / 3.0


def style_similarity(a: torch.Tensor, b: torch.Tensor, model) -> float:
    """style_loss_fn of eval_synthetic.py:143-156: mean over the embedding dict of cosine_similarity (the similarity itself)."""
    from st_ito.utils import get_param_embeds

    ea, eb = get_param_embeds(a, model, 48000), get_param_embeds(b, model, 48000)
    return float(torch.stack([torch.nn.functional.cosine_similarity(ea[k], eb[k]) for k in ea]).mean())


def load_examples(input_dir: str, fade_samples: int = 32768):
    """eval_synthetic.py:263-299: dry examples (faded in) and, per test case, the test examples -- which the reference fades into
    a variable it never uses (`test_audio_fade`), so in effect apply_fade_in's IN-PLACE multiplication is what fades them too."""
    from st_ito.audio_io import load_wav, resample
    from st_ito.utils import apply_fade_in

    def read(path):
        x, sr = load_wav(path)
        if sr != 48000:
            x = resample(x, sr, 48000)
        return apply_fade_in(x, fade_samples).unsqueeze(0)

    dry = {"music": {}, "speech": {}, "vocals": {}}
    for path in sorted(glob.glob(os.path.join(input_dir, "dry", "*.wav"))):
        name = os.path.basename(path).replace(".wav", "")
        dry[get_source_type(name)][name] = read(path)
    tests = {}
    for case in TEST_CASES:
        files = sorted(glob.glob(os.path.join(input_dir, case, "*.wav")))
        if not files:
            continue
        tests[case] = {"music": {}, "speech": {}, "vocals": {}}
        for path in files:
            name = os.path.basename(path).replace(".wav", "")
            tests[case][get_source_type(name)][name] = read(path)
    return dry, tests


def finish_triplet(output_audio: torch.Tensor, test_audio: torch.Tensor, gt_audio: torch.Tensor):
    """eval_synthetic.py:396-425: batch dimension off, all three cropped to their common length, each scaled to -22 LUFS."""
    from st_ito.loudness import normalize_loudness

    out, tst, gt = output_audio.squeeze(0), test_audio.squeeze(0), gt_audio.squeeze(0)
    n = min(out.shape[-1], tst.shape[-1], gt.shape[-1])
    return tuple(normalize_loudness(a[..., :n].cpu(), 48000, -22.0)[0] for a in (out, tst, gt))


def run_example(dry_audio, test_audio, gt_audio, plugins_spec, model, max_iters: int = 32, popsize: int = 128, seed=None):
    """One (dry example, test case) of the loop for the ES method on the pb set (eval_synthetic.py:338-394).  The harness hands
    run_es the SAME tensors every method sees; run_es peak-normalises them in place, so the errors are computed on what it left
    (as in the reference).  -> (result row, (output, target, ground truth) as saved)"""
    import copy
    from st_ito.style_transfer import load_plugins, run_es
    from st_ito.utils import get_param_embeds

    plugins, _, _ = load_plugins(copy.deepcopy(plugins_spec))              # reloaded per example (341-345)
    t0 = time.time()
    result = run_es(dry_audio, test_audio, 48000, plugins, model, get_param_embeds, normalization="peak", max_iters=max_iters,
                    sigma0=0.33, distance="cosine", popsize=popsize, dropout=0.0, save_pop=False, find_w0=False, random_crop=False,
                    w0=None, seed=seed)
    elapsed = time.time() - t0
    out = result["output_audio"]
    out = out.unsqueeze(0) if out.ndim == 2 else out
    n = min(out.shape[-1], gt_audio.shape[-1])                             # (the reference relies on equal lengths; guarded here)
    o, g = out[..., :n], gt_audio[..., :n]
    row = {"elapsed_time": elapsed,
           "mrstft_error": float(mrstft_error(o, g)),
           "mrstft_error_norm": float(mrstft_error(o / o.abs().max(), g / g.abs().max())),
           "style_error_gt": style_similarity(o.clone(), g.clone(), model),
           "style_error_target": style_similarity(out.clone(), test_audio.clone(), model)}
    return row, finish_triplet(out, test_audio, gt_audio), result


def run_synthetic_benchmark(input_dir: str, output_dir: str, model, max_iters: int = 32, popsize: int = 128, seed=None,
                            fade_samples: int = 32768):
    from st_ito.audio_io import save_wav

    rng = np.random.RandomState(seed) if seed is not None else np.random
    os.makedirs(output_dir, exist_ok=True)
    dry, tests = load_examples(input_dir, fade_samples)
    spec = get_pb_plugins()
    results, n_ex = {}, 0
    for source_type, dry_examples in dry.items():
        for dry_name, dry_audio in dry_examples.items():
            for case, by_type in tests.items():
                names = list(by_type[source_type].keys())
                if dry_name not in by_type[source_type] or len(names) < 2:
                    continue
                test_name = dry_name
                while test_name == dry_name:                                # another example of this source type (317-321)
                    test_name = names[int(rng.randint(len(names)))]
                test_audio = by_type[source_type][test_name]
                gt_audio = by_type[source_type][dry_name]
                example_id = f"{dry_name}->{case}-{test_name}"
                print(f"Processing {example_id}")
                row, (o, t, g), _ = run_example(dry_audio, test_audio, gt_audio, spec, model, max_iters, popsize,
                                                None if seed is None else seed + n_ex)
                n_ex += 1
                results.setdefault(case, {})[example_id] = {"style-es (param-panns)_pb": row}
                ex_dir = os.path.join(output_dir, example_id)
                os.makedirs(ex_dir, exist_ok=True)
                save_wav(os.path.join(ex_dir, f"{example_id}_style-es (param-panns)_pb.wav"), o, 48000)
                save_wav(os.path.join(ex_dir, f"{example_id}_target.wav"), t, 48000)
                save_wav(os.path.join(ex_dir, f"{example_id}_gt.wav"), g, 48000)
                with open(os.path.join(output_dir, "results.json"), "w") as fp:
                    json.dump(results, fp, indent=2)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("input_dir", type=str)
    ap.add_argument("--output_dir", type=str, default=os.path.join("output", "synthetic"))
    ap.add_argument("--fade_samples", type=int, default=32768)
    ap.add_argument("--max-iters", type=int, default=32)
    ap.add_argument("--popsize", type=int, default=128)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--ckpt", default=None, help="AFx-Rep checkpoint; omitted: seeded random weights")
    a = ap.parse_args(argv)
    from st_ito.utils import load_param_model, make_synthetic_param_model

    model = load_param_model(a.ckpt, use_gpu=True) if a.ckpt else make_synthetic_param_model(0)
    return run_synthetic_benchmark(a.input_dir, a.output_dir, model, a.max_iters, a.popsize, a.seed, a.fade_samples)


if __name__ == "__main__":
    main()
