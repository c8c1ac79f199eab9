#!/usr/bin/env python
"""Parameter-recovery case study on the MI355X path -- the ES arm of the reference's scripts/eval/eval_case_study.py
(plugin cases 226-344, study loop 346-522): one plugin with all parameters but one FIXED, the free parameter stepped over its
range; for every step a target is rendered with the known value (process_audio), run_es estimates it back from the
(input, target) pair with the harness's settings (popsize 128, sigma0 0.33, find_w0 False, random_crop True, 478-494), and
(estimated raw value, fopt) is recorded per (mode, method, parameter, value) like the reference's JSON.

Differences, all stated: only the `pb_*` cases (Basic* effects) and the `param-panns` method are built -- the `vst_*` cases need
binary plugins, `clap` another model; the audio comes from `--audio files...` or `--synthetic` (the reference walks its own
dataset directories); `--seed` seeds the draws and the CMA-ES (the reference seeds Lightning once and leaves pycma unseeded).

    python st-ito_amd/scripts/eval_case_study.py --plugins pb_Compressor pb_Reverb --synthetic --num-steps 4 --num-runs 2 --max-iters 5
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

MIN_LEN = 524288 + 48000   # eval_case_study.py:403: a source must be long enough for the longest crop + both guards


def get_case(plugin_name: str):
    """(plugins, parameter_under_test, min_param_value, max_param_value) of eval_case_study.py:226-344.  The values of
    fixed_parameters are in the parameters' OWN units (Parameter.set_value); the "our_bypass" entry the reference lists among
    them is never looked at (process_audio treats that slot before it consults fixed_parameters) and is kept for fidelity."""
    from st_ito.effects import BasicChorus, BasicCompressor, BasicDelay, BasicDistortion, BasicParametricEQ, BasicReverb

    def one(cls, nch, fixed):
        return {plugin_name: {"class_path": cls, "num_params": None, "num_channels": nch, "fixed_parameters": dict(fixed, our_bypass=0.0)}}

    if plugin_name == "pb_ParametricEQ":
        fixed = {"low_shelf_cutoff_freq": 120.0, "low_shelf_q_factor": 0.707, "high_shelf_gain_db": 0.0,
                 "high_shelf_cutoff_freq": 10000.0, "high_shelf_q_factor": 0.707}
        for b, fc in enumerate((300.0, 1000.0, 3000.0, 10000.0)):
            fixed.update({f"band{b}_gain_db": 0.0, f"band{b}_cutoff_freq": fc, f"band{b}_q_factor": 0.707})
        return one(BasicParametricEQ, 1, fixed), "low_shelf_gain_db", 0.0, 1.0
    if plugin_name == "pb_Chorus":
        return one(BasicChorus, 1, {"rate_hz": 1.0, "centre_delay_ms": 7.0, "depth": 0.1, "feedback": 0.5}), "mix", 0.0, 1.0
    if plugin_name == "pb_Compressor":
        return one(BasicCompressor, 1, {"ratio": 4.0, "attack_ms": 1.0, "release_ms": 100.0}), "threshold_db", 0.0, 1.0
    if plugin_name == "pb_Distortion":
        return one(BasicDistortion, 1, {"output_gain_db": 0.0}), "drive_db", 0.5, 1.0
    if plugin_name == "pb_Delay":
        return one(BasicDelay, 2, {"delay_seconds": 0.1, "feedback": 0.5}), "mix", 0.0, 1.0
    if plugin_name == "pb_Reverb":
        return one(BasicReverb, 2, {"damping": 0.4, "wet_dry": 0.4, "width": 0.8}), "room_size", 0.0, 1.0
    if plugin_name.startswith("vst_"):
        raise NotImplementedError("VST plugins (pedalboard.load_plugin) are not supported in this build")
    raise ValueError(f"Unknown plugin_name: {plugin_name}")


def crop_pair(input_audio: torch.Tensor, target_audio: torch.Tensor, rng):
    """eval_case_study.py:432-455: two crop lengths in [262144, 524288), then a start for each at least 48000 samples from
    either end, peak normalisation of each crop, mono -> stereo by repetition.  The draws come in the reference's order."""
    input_len = int(rng.randint(262144, 524288))
    target_len = int(rng.randint(262144, 524288))
    start = int(rng.randint(48000, input_audio.shape[1] - input_len - 48000))
    x = input_audio[:, start:start + input_len].clone()
    x /= x.abs().max()
    start = int(rng.randint(48000, target_audio.shape[1] - target_len - 48000))
    t = target_audio[:, start:start + target_len].clone()
    t /= t.abs().max()
    if x.shape[0] == 1:
        x = x.repeat(2, 1)
    if t.shape[0] == 1:
        t = t.repeat(2, 1)
    return x, t


def study_point(plugins_spec, plugin_name: str, parameter_under_test: str, parameter_value: float, pick, model, rng,
                max_iters: int = 5, popsize: int = 128, seed=None):
    """One (value, run) of the study (eval_case_study.py:358-494).  pick(rng) -> (input_audio, target_audio), called where the
    reference draws its files (after the dummy render).  -> dict(estimated_param, fopt, target_value, wopt, ...)"""
    import copy
    from st_ito.style_transfer import load_plugins, process_audio, run_es
    from st_ito.utils import get_param_embeds

    plugins, _, init_params = load_plugins(copy.deepcopy(plugins_spec))      # reloaded for every run (358-360)
    idx = plugins[plugin_name]["parameter_names"].index(parameter_under_test)
    test_init_params = list(init_params)
    test_init_params[idx] = parameter_value
    # the dummy call of 372-375: it leaves the values in the plugin instance, from which the study reads the target back
    process_audio(rng.randn(2, 131072).astype(np.float32), np.asarray(test_init_params), 48000, plugins)
    prm = plugins[plugin_name]["instance"].parameters[parameter_under_test]
    target_value = prm.get_value() if hasattr(prm, "get_value") else prm.raw_value
    input_audio, target_audio = pick(rng)
    x, t = crop_pair(input_audio, target_audio, rng)
    audio_output = torch.from_numpy(process_audio(t.numpy(), np.asarray(test_init_params), 48000, plugins))
    result = run_es(x.unsqueeze(0), audio_output.unsqueeze(0), 48000, plugins, model, get_param_embeds, max_iters=max_iters, w0=None,
                    find_w0=False, sigma0=0.33, distance="cosine", random_crop=True, popsize=popsize, parallel=False, dropout=0.0,
                    seed=seed)
    return {"estimated_param": float(result["wopt"][idx]), "fopt": float(result["fopt"]), "target_value": float(target_value),
            "wopt": result["wopt"], "params": result["params"], "target_audio": audio_output, "output_audio": result["output_audio"]}


def run_case_study(plugin_names, sources, model, out_dir: str, num_runs: int = 3, num_steps: int = 4, max_iters: int = 5,
                   popsize: int = 128, mode: str = "different", seed=None, save_audio: bool = False):
    """sources: list of (audio (chs, n) float32 at 48 kHz) with n >= 524288 + 48000.  Writes <out_dir>/<plugin>/
    case_study_results.json in the reference's shape: results[mode][method][parameter][value] = [(estimate, fopt), ...]."""
    from st_ito.audio_io import save_wav

    rng = np.random.RandomState(seed) if seed is not None else np.random
    sources = [s for s in sources if s.shape[-1] >= MIN_LEN]
    if not sources:
        raise ValueError(f"no source of at least {MIN_LEN} samples (eval_case_study.py:403)")
    all_results = {}
    for plugin_name in plugin_names:
        spec, parameter_under_test, lo, hi = get_case(plugin_name)
        pdir = os.path.join(out_dir, plugin_name)
        os.makedirs(os.path.join(pdir, "audio"), exist_ok=True)
        results = {mode: {"param-panns": {parameter_under_test: {}}}}
        n_point = 0
        for parameter_value in np.linspace(lo, hi, num_steps):
            runs = results[mode]["param-panns"][parameter_under_test].setdefault(float(parameter_value), [])
            for n in range(num_runs):
                def pick(r):   # 394-420: a source for the input and, in mode "different", another draw for the target
                    a_in = sources[int(r.randint(len(sources)))]
                    return a_in, (sources[int(r.randint(len(sources)))] if mode == "different" else a_in)

                pt = study_point(spec, plugin_name, parameter_under_test, float(parameter_value), pick, model, rng,
                                 max_iters=max_iters, popsize=popsize, seed=None if seed is None else seed + n_point)
                n_point += 1
                runs.append((pt["estimated_param"], pt["fopt"]))
                print(f"{plugin_name} {parameter_under_test} = {parameter_value:0.2f} (run {n + 1}/{num_runs}): estimated {pt['estimated_param']:0.3f}, fopt {pt['fopt']:0.4f}")
                if save_audio:
                    save_wav(os.path.join(pdir, "audio", f"{parameter_under_test}_{parameter_value:0.2f}_{n}.wav"), pt["target_audio"], 48000)
                    save_wav(os.path.join(pdir, "audio", f"{parameter_under_test}_{parameter_value:0.2f}_{n}_param-panns_{pt['estimated_param']:0.2f}.wav"),
                             pt["output_audio"], 48000)
                with open(os.path.join(pdir, "case_study_results.json"), "w") as f:
                    json.dump(results, f, indent=4)
        all_results[plugin_name] = results
    return all_results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--plugins", nargs="+", default=["pb_Distortion", "pb_Compressor", "pb_ParametricEQ"])   # the reference's active list (68-82)
    ap.add_argument("--audio", nargs="*", default=[], help="wav files (any rate: resampled to 48 kHz)")
    ap.add_argument("--synthetic", action="store_true", help="two seeded synthetic sources instead of files")
    ap.add_argument("--num-runs", type=int, default=3)
    ap.add_argument("--num-steps", type=int, default=4)
    ap.add_argument("--max-iters", type=int, default=5)
    ap.add_argument("--popsize", type=int, default=128)
    ap.add_argument("--mode", default="different", choices=["different", "same"])
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--save-audio", action="store_true")
    ap.add_argument("--ckpt", default=None, help="AFx-Rep checkpoint; omitted: seeded random weights")
    ap.add_argument("--output-dir", default=os.path.join("output", "new_case_study"))
    a = ap.parse_args(argv)

    from st_ito.audio_io import load_wav, resample
    from st_ito.utils import load_param_model, make_synthetic_param_model

    model = load_param_model(a.ckpt, use_gpu=True) if a.ckpt else make_synthetic_param_model(0)
    sources = []
    if a.synthetic:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
        from bench import synth_audio
        sources = [synth_audio(700 + i, 2, MIN_LEN + 48000) for i in range(2)]
    for path in a.audio:
        x, sr = load_wav(path)
        sources.append(resample(x, sr, 48000) if sr != 48000 else x)
    if not sources:
        ap.error("give --audio files or --synthetic")
    return run_case_study(a.plugins, sources, model, a.output_dir, a.num_runs, a.num_steps, a.max_iters, a.popsize, a.mode, a.seed, a.save_audio)


if __name__ == "__main__":
    main()
