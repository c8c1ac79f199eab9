#!/usr/bin/env python
"""Production-style-transfer benchmark harness on the MI355X path -- the ES arm of the reference's
scripts/eval/eval_pst.py (get_plugins 206-650 for the pedalboard chains, run_pst_benchmark 652-903,
method settings 974-991: popsize 128, 32 iterations, sigma0 0.33, find_w0 False, random_crop True).

Differences, all stated: only the `*-pb` chains (Basic* effects) and the `style-es` method are
built -- the VST chains need binary plugins and the other methods (random, rule-based, DeepAFx-ST)
other models.  The examples come from `--pairs file` (one "input.wav<TAB>target.wav" per line,
relative to --root-dir) or `--synthetic N`; the reference's hard-coded file lists are its own
dataset and are not reproduced.  With `--batched` all examples of equal length are optimised together
by run_es_batch (BASELINE.json configs[2]) instead of one after the other.

    python st-ito_amd/scripts/eval_pst.py --chain general-pb --synthetic 4 --max-iters 8 --popsize 32
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def get_plugins(chain_type: str):
    """The pedalboard ("-pb") chains of eval_pst.py:269-301, 559-645.  guitar-pb lists
    "ParametricEQ" twice in one dict literal there, so the second entry replaces the first in
    place: the chain really is Compressor, ParametricEQ, Distortion, Reverb."""
    from st_ito.effects import BasicCompressor, BasicDelay, BasicDistortion, BasicParametricEQ, BasicReverb

    one = lambda cls: {"class_path": cls, "num_params": None, "num_channels": 1, "fixed_parameters": {}}
    two = lambda cls: {"class_path": cls, "num_params": None, "num_channels": 2, "fixed_parameters": {}}
    if chain_type == "general-pb":
        return OrderedDict(Distortion=one(BasicDistortion), ParametricEQ=one(BasicParametricEQ),
                           Compressor=one(BasicCompressor), Delay=two(BasicDelay), Reverb=two(BasicReverb))
    elif chain_type == "mastering-pb":
        return OrderedDict(ParametricEQ=one(BasicParametricEQ), Compressor=one(BasicCompressor), Reverb=two(BasicReverb))
    elif chain_type == "vocals-pb":
        return OrderedDict(ParametricEQ=one(BasicParametricEQ), Compressor=one(BasicCompressor),
                           Distortion=one(BasicDistortion), Delay=two(BasicDelay), Reverb=two(BasicReverb))
    elif chain_type == "guitar-pb":
        return OrderedDict(Compressor=one(BasicCompressor), ParametricEQ=one(BasicParametricEQ),
                           Distortion=one(BasicDistortion), Reverb=two(BasicReverb))
    raise ValueError(f"Unknown chain_type: {chain_type}")


def prepare_pair(input_audio: torch.Tensor, input_sr: int, target_audio: torch.Tensor, target_sr: int,
                 fade_samples: int = 32768):
    """eval_pst.py:705-748: resample to 48 kHz, make stereo, add the batch dim, fade in."""
    from st_ito.audio_io import resample
    from st_ito.utils import apply_fade_in

    if input_sr != 48000:
        input_audio = resample(input_audio, input_sr, 48000)
    if target_sr != 48000:
        target_audio = resample(target_audio, target_sr, 48000)
    min_len = min(input_audio.shape[1], target_audio.shape[1], 262144)
    if input_audio.shape[0] == 1:
        input_audio = input_audio.repeat(2, 1)
    if target_audio.shape[0] == 1:
        target_audio = target_audio.repeat(2, 1)
    # apply_fade_in works in place (utils.py:31-43); the caller's tensors are left alone
    return (apply_fade_in(input_audio.unsqueeze(0).clone(), fade_samples),
            apply_fade_in(target_audio.unsqueeze(0).clone(), fade_samples), min_len)


def style_distance(a: torch.Tensor, b: torch.Tensor, model, sr: int) -> float:
    """Mean over {mid, side} of cosine_similarity (the harness reports the similarity itself,
    eval_pst.py:811-836)."""
    from st_ito.utils import get_param_embeds

    ea, eb = get_param_embeds(a, model, sr), get_param_embeds(b, model, sr)
    return float(torch.stack([torch.cosine_similarity(ea[k], eb[k], dim=1) for k in ea]).mean())


def run_pst_benchmark(pairs, plugins, model, out_dir: str, max_iters: int = 32, popsize: int = 128, sigma0: float = 0.33,
                      random_crop: bool = True, seed: int = None, batched: bool = False, tag: str = "pb"):
    """pairs: list of (name, input (chs, n), input_sr, target (chs, n), target_sr).  Returns the results dict
    the reference dumps to JSON: per method, per metric, one value per example (+ time_elapsed)."""
    from st_ito.audio_io import save_wav
    from st_ito.loudness import normalize_loudness
    from st_ito.style_transfer import load_plugins, run_es, run_es_batch
    from st_ito.utils import get_param_embeds

    os.makedirs(out_dir, exist_ok=True)
    plugins, _, _ = load_plugins(plugins)
    sr = 48000
    prepared = [(name,) + prepare_pair(x, xsr, t, tsr) for name, x, xsr, t, tsr in pairs]
    results = {m: {"time_elapsed": [], "style_features": []} for m in ("input", "style-es (param-panns)")}

    es_out = [None] * len(prepared)
    if batched and len({(p[1].shape, p[2].shape) for p in prepared}) == 1 and prepared[0][1].shape == prepared[0][2].shape:
        t0 = time.time()
        res = run_es_batch(torch.cat([p[1] for p in prepared]), torch.cat([p[2] for p in prepared]), sr, plugins, model,
                           get_param_embeds, max_iters=max_iters, sigma0=sigma0, popsize=popsize, random_crop=random_crop, seed=seed)
        dt = (time.time() - t0) / len(prepared)
        es_out = [(r, dt) for r in res]
    for idx, (name, xin, tgt, min_len) in enumerate(prepared):
        if es_out[idx] is None:
            t0 = time.time()
            r = run_es(xin.clone(), tgt.clone(), sr, plugins, model, get_param_embeds, max_iters=max_iters, sigma0=sigma0,
                       popsize=popsize, find_w0=False, random_crop=random_crop, distance="cosine", dropout=0.0,
                       seed=None if seed is None else seed + idx)
            es_out[idx] = (r, time.time() - t0)
        for method, (out_audio, elapsed, params) in {
            "input": (xin[0], 0.0, None),
            "style-es (param-panns)": (es_out[idx][0]["output_audio"], es_out[idx][1], es_out[idx][0]["params"]),
        }.items():
            results[method]["time_elapsed"].append(elapsed)
            results[method]["style_features"].append(style_distance(out_audio[None], tgt, model, sr))
            stem = f"{idx:02d}_{method.split(' ')[0]}_{tag}"
            if params is not None:
                with open(os.path.join(out_dir, stem + ".json"), "w") as f:
                    json.dump(params, f, indent=2)
            out_audio, _ = normalize_loudness(out_audio[:, :min_len], sr, -22.0)   # eval_pst.py:843-853
            save_wav(os.path.join(out_dir, stem + ".wav"), out_audio, sr)
        tgt_n, _ = normalize_loudness(tgt[0][:, :min_len], sr, -22.0)
        save_wav(os.path.join(out_dir, f"{idx:02d}_target_{tag}.wav"), tgt_n, sr)
    print()
    for method, res in results.items():
        print(f"{method}:\n\tstyle_features: {np.mean(res['style_features'])}")
    with open(os.path.join(out_dir, f"results_{datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}.json"), "w") as f:
        json.dump(results, f, indent=2)
    return results


def synthetic_pairs(n: int, seconds: float, plugins):
    """Inputs: seeded noise + tones (the bench's recipe); targets: another signal through the same
    chain at random parameters, so that a perfect match exists in the search space."""
    from st_ito.style_transfer import load_plugins, process_audio
    import copy

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from bench import synth_audio

    pl, D, _ = load_plugins(copy.deepcopy(plugins))
    ns = int(seconds * 48000)
    out = []
    for i in range(n):
        w = np.random.default_rng(100 + i).random(D) * 0.5  # bypass slots stay < 0.5 (they are dead anyway)
        tgt = torch.from_numpy(process_audio(synth_audio(500 + i, 2, ns).numpy(), w, 48000, pl))
        out.append((f"synthetic{i}", synth_audio(400 + i, 2, ns), 48000, tgt, 48000))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--chain", default="general-pb", choices=["general-pb", "mastering-pb", "vocals-pb", "guitar-pb"])
    ap.add_argument("--pairs", help="text file: input<TAB>target per line")
    ap.add_argument("--root-dir", default=".")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic pairs instead of files")
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--max-iters", type=int, default=32)
    ap.add_argument("--popsize", type=int, default=128)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--batched", action="store_true")
    ap.add_argument("--ckpt", default=None, help="AFx-Rep checkpoint; omitted: seeded random weights")
    ap.add_argument("--output-dir", default=os.path.join("output", "pst"))
    a = ap.parse_args(argv)

    from st_ito.audio_io import load_wav
    from st_ito.utils import load_param_model, make_synthetic_param_model

    model = load_param_model(a.ckpt, use_gpu=True) if a.ckpt else make_synthetic_param_model(0)
    plugins = get_plugins(a.chain)
    if a.synthetic:
        pairs = synthetic_pairs(a.synthetic, a.seconds, plugins)
    elif a.pairs:
        pairs = []
        for line in open(a.pairs):
            if line.strip():
                i, t = line.rstrip("\n").split("\t")
                (x, xsr), (y, ysr) = load_wav(os.path.join(a.root_dir, i)), load_wav(os.path.join(a.root_dir, t))
                pairs.append((os.path.basename(i), x, xsr, y, ysr))
    else:
        ap.error("give --pairs or --synthetic N")
    return run_pst_benchmark(pairs, plugins, model, os.path.join(a.output_dir, a.chain), a.max_iters, a.popsize,
                             seed=a.seed, batched=a.batched, tag=a.chain)


if __name__ == "__main__":
    main()
