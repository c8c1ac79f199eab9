#!/usr/bin/env python
"""Randomised parity soak: random effect chains (with repeats), channel counts, lengths (odd ones included),
population sizes, bypass slots, fixed parameters and per-stage normalisation, rendered on the GPU and compared with
the oracle candidate by candidate.  A flagged case is not necessarily a defect: every Distortion multiplies
differences by up to 10^(48/20) = 251 at zero crossings, so chains with several of them are ill-conditioned
(tools/soak_case.py replays one case prefix by prefix: seed 2 case 45 goes 2e-7, 2e-7, 4e-6, 3e-4 through
Distortion+Delay+Distortion+Distortion).  tests/test_gpu_soak.py runs 40 seeded cases of >= 3 stages inside the suite; for longer hunts on a GPU box:
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


def draw_case(rng, case, min_fx=1):
    """One random render case from the generator's stream (the order of the draws is the file format of profiles/round5_soak.txt:
    seed + case number reproduce a case; min_fx = 1 is the stream those files were taken with)."""
    n_fx = int(rng.integers(min_fx, 6))
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
    return dict(kinds=kinds, with_bypass=with_bypass, ns=ns, chs=chs, n=n, P=P, op=op, pp=pp, x=x, W=W)


def render_case(c, dev):
    """-> largest |HIP - oracle| over the candidates of the case, on peak-normalised audio (i.e. as a fraction of the peak)."""
    audio, peaks = engine.render_population(c["pp"], torch.from_numpy(c["x"]).to(dev), torch.from_numpy(c["W"]).to(dev), SR,
                                            chain=engine.compile_chain(c["pp"], c["ns"]))
    engine.normalize_audio_(audio, peaks)
    got = audio.cpu().numpy()
    err = 0.0
    for p in range(c["P"]):
        ref = O.process_audio(c["x"].copy(), c["W"][p], SR, c["op"], normalize_stages=c["ns"])
        assert ref.shape == got[p].shape, (c["kinds"], ref.shape, got[p].shape)
        err = max(err, float(np.abs(got[p] - ref).max()))
    return err


def describe(c):
    return f"{'+'.join(c['kinds']):60s} chs={c['chs']} n={c['n']:6d} P={c['P']} bypass={int(c['with_bypass'])} ns={int(c['ns'])}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--min-fx", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda", 0)
    worst = 0.0
    for case in range(a.cases):
        c = draw_case(rng, case, a.min_fx)
        err = render_case(c, dev)
        worst = max(worst, err)
        flag = "" if err < 1e-4 else "   <-- CHECK"
        print(f"case {case:3d}: {describe(c)}  max err {err:.2e}{flag}", flush=True)
    print(f"worst error over {a.cases} cases: {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
