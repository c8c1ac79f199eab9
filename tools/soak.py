#!/usr/bin/env python
"""Randomised parity soak: random effect chains (with repeats), channel counts, lengths (odd ones included),
population sizes, bypass slots, fixed parameters and per-stage normalisation, rendered on the GPU and compared with
the oracle candidate by candidate.  A flagged case is not necessarily a defect: every Distortion multiplies
differences by up to 10^(48/20) = 251 at zero crossings, so chains with several of them are ill-conditioned
(tools/soak_case.py replays one case prefix by prefix: seed 2 case 45 goes 2e-7, 2e-7, 4e-6, 3e-4 through
Distortion+Delay+Distortion+Distortion).  Not part of the test suite (the oracle side is slow); run it on a GPU box:
    python tools/soak.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import st_ito_oracle as O
from st_ito import effects as E, engine

SR = 48000
KINDS = {"ParametricEQ": (E.BasicParametricEQ, 1), "Compressor": (E.BasicCompressor, 1), "Distortion": (E.BasicDistortion, 1),
         "Delay": (E.BasicDelay, 2), "Reverb": (E.BasicReverb, 2), "Gain": (E.BasicGain, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda", 0)
    worst = 0.0
    for case in range(a.cases):
        n_fx = int(rng.integers(1, 6))
        kinds = [str(k) for k in rng.choice(list(KINDS), n_fx)]
        with_bypass = bool(rng.integers(0, 2))
        ns = bool(rng.integers(0, 2))
        chs = int(rng.integers(1, 3))
        n = int(rng.choice([1, 5, 191, 193, 4096, 4097, 30011, 48000, 65536, 100003]))
        P = int(rng.integers(1, 5))
        op = O.make_plugins(kinds, with_bypass)
        spec, seen = [], {}
        for k in kinds:
            seen[k] = seen.get(k, 0) + 1
            spec.append((k if seen[k] == 1 else f"{k}{seen[k]}", KINDS[k][0], KINDS[k][1]))
        pp = E.make_plugins(spec, with_bypass)
        if rng.integers(0, 3) == 0 and "Compressor" in op:   # a fixed parameter now and then
            op["Compressor"]["fixed_parameters"] = {"ratio": 6.0}
            pp["Compressor"]["fixed_parameters"] = {"ratio": 6.0}
        D = sum(p["num_params"] for p in op.values())
        x = (O.synth_audio(1000 + case, chs, max(n, 2))[:, :n] * float(rng.uniform(0.05, 1.0))).numpy()
        W = rng.random((P, D))
        audio, peaks = engine.render_population(pp, torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), SR,
                                                chain=engine.compile_chain(pp, ns))
        engine.normalize_audio_(audio, peaks)
        got = audio.cpu().numpy()
        err = 0.0
        for p in range(P):
            ref = O.process_audio(x.copy(), W[p], SR, op, normalize_stages=ns)
            assert ref.shape == got[p].shape, (kinds, ref.shape, got[p].shape)
            err = max(err, float(np.abs(got[p] - ref).max()))
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- CHECK"
        print(f"case {case:3d}: {'+'.join(kinds):60s} chs={chs} n={n:6d} P={P} bypass={int(with_bypass)} ns={int(ns)}  max err {err:.2e}{flag}", flush=True)
    print(f"worst error over {a.cases} cases: {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
