#!/usr/bin/env python
"""The five BASELINE.json configs at the shape ONE GPU sees (multi-GPU configs: the per-rank share),
timed like bench.py (ask -> evaluate -> tell, inputs resident in HBM, synthetic audio, seeded random
AFx-Rep weights).  bench.py itself only runs configs[1]; this tool is the evidence that the other
shapes run and what they cost.
    python tools/run_configs.py [--steps 3] [--only 1,2,3,4,5,6,7]   (6, 7: the reference's own operating points, whole run_es calls)"""
import argparse
import functools
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import synth_audio
from st_ito import cmaes, effects as E
from st_ito.engine import PopulationEvaluator
from st_ito.style_transfer import process_audio
from st_ito.utils import get_param_embeds, make_synthetic_param_model

SR = 48000


def plugins_of(spec):
    pl = {}
    for name, cls, nch in spec:
        inst = cls()
        names = list(inst.parameters.keys())
        pl[name] = {"class_path": cls, "num_params": len(names), "num_channels": nch, "fixed_parameters": {},
                    "instance": inst, "parameter_names": names}
    return pl


def run(label, spec, chs, seconds, pop, pairs, steps, model):
    n = int(seconds * SR)
    pl = plugins_of(spec)
    D = sum(p["num_params"] for p in pl.values())
    xs = torch.stack([synth_audio(100 + b, chs, n) for b in range(pairs)])
    tg = torch.stack([torch.from_numpy(process_audio(synth_audio(200 + b, chs, n).numpy(), np.random.default_rng(b).random(D), SR, pl))
                      for b in range(pairs)])
    te = get_param_embeds(tg, model, SR)
    ev = PopulationEvaluator(xs, SR, pl, model, te)
    ess = [cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": pop, "seed": 42 + b}) for b in range(pairs)]

    def step():
        Ws = [es.ask() for es in ess]
        f = ev.evaluate(np.concatenate([np.asarray(W) for W in Ws], 0))[0].tolist()
        for b, es in enumerate(ess):
            es.tell(Ws[b], f[b * pop:(b + 1) * pop])

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{label:58s} D={D:3d}  {pairs * pop:5d} cand/step  {dt * 1e3:9.1f} ms/step  {pairs * pop / dt:9.1f} cand/s  "
          f"({pairs * pop * seconds / dt:8.0f} audio-seconds/s)", flush=True)


def run_es_point(label, plugins, pop, iters, seconds, random_crop, find_w0, model):
    """One whole run_es call (the reference's user path: style_transfer.py:399-692) at one of the reference's own operating
    points, timed from the call to the result: candidate evaluations (the find_w0 batch included) / wall time."""
    from st_ito.style_transfer import load_plugins, run_es
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # load_plugins prints every parameter, like the reference
        plugins, D, _ = load_plugins(plugins)
    n = int(seconds * SR)
    x = synth_audio(300, 2, n)[None]
    tg = torch.from_numpy(process_audio(synth_audio(301, 2, n).numpy(), np.random.default_rng(3).random(D), SR, plugins))[None]
    kw = dict(max_iters=iters, popsize=pop, sigma0=0.33, random_crop=random_crop, find_w0=find_w0, seed=11, early_stop=False)
    run_es(x.clone(), tg.clone(), SR, plugins, model, get_param_embeds, **dict(kw, max_iters=2))   # warm (packing, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = run_es(x.clone(), tg.clone(), SR, plugins, model, get_param_embeds, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evals = pop * (len(r["fval_history"]) - 1) + (pop if find_w0 else 0)
    print(f"{label:58s} D={D:3d}  {pop:5d} cand/step  {dt * 1e3 / max(len(r['fval_history']) - 1, 1):9.1f} ms/iter  {evals / dt:9.1f} cand/s  "
          f"(run_es: {evals} evaluations of 262144 samples in {dt:.2f} s, fopt {r['fopt']:.4f})", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default="1,2,3,4,5,6,7")
    a = ap.parse_args()
    only = {int(v) for v in a.only.split(",")}
    model = make_synthetic_param_model(seed=0, input_norm="minmax")
    eq, comp, rev, gain = E.BasicParametricEQ, E.BasicCompressor, E.BasicReverb, E.BasicGain
    five = [("ParametricEQ", eq, 1), ("Compressor", comp, 1), ("Reverb", rev, 2), ("ParametricEQ2", eq, 1), ("Gain", gain, 1)]
    conv = [("ParametricEQ", eq, 1), ("Compressor", comp, 1), ("ConvReverb", functools.partial(E.NoiseShapedReverb, num_samples=96000), 2),
            ("ParametricEQ2", eq, 1), ("Gain", gain, 1)]
    cfgs = {
        1: ("configs[0] mono 2 s, EQ+comp, pop 8", [("ParametricEQ", eq, 1), ("Compressor", comp, 1)], 1, 2.0, 8, 1),
        2: ("configs[1] stereo 10 s, 5-effect chain, pop 256", five, 2, 10.0, 256, 1),
        3: ("configs[2] 16 pairs x pop 128, stereo 10 s, 5-effect chain", five, 2, 10.0, 128, 16),
        4: ("configs[3] per-GPU share: pop 256, stereo 30 s", five, 2, 30.0, 256, 1),
        5: ("configs[4] per-GPU share: pop 128, stereo 30 s, conv reverb 96000 taps", conv, 2, 30.0, 128, 1),
    }
    for k in sorted(only):
        if k in cfgs:
            label, spec, chs, sec, pop, pairs = cfgs[k]
            run(label, spec, chs, sec, pop, pairs, a.steps, model)
    # the reference's own operating points (not BASELINE configs): scripts/eval/eval_pst.py:974-991 (pop 128, 262144-sample random
    # crops, general-pb chain, 32 iterations per example) and the CLI default scripts/run_optim.py:304-306 (pop 32, basic chain,
    # find_w0, input padded / cropped to 262144 samples)
    if 6 in only:
        sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
        from eval_pst import get_plugins
        run_es_point("PST harness: pop 128, 10 s stereo -> random crop 262144, general-pb", get_plugins("general-pb"), 128, 16, 10.0, True, False, model)
    if 7 in only:
        run_es_point("CLI default: pop 32, 5 s stereo padded to 262144, basic chain, find_w0", E.make_plugins("basic"), 32, 24, 5.0, False, True, model)


if __name__ == "__main__":
    main()
