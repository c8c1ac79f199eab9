#!/usr/bin/env python
"""Randomised parity soak of the whole evaluate step (render -> log-mel -> Cnn14 -> cosine loss): random chains,
channel counts, lengths and population sizes; per-candidate losses against the oracle's evaluate.
    python tools/soak_eval.py [--cases 20] [--seed 0]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import st_ito_oracle as O
from st_ito import effects as E
from st_ito.engine import PopulationEvaluator
from st_ito.models.panns import Cnn14
from st_ito.utils import get_param_embeds
from soak import KINDS, SR


def draw_eval_case(rng, case):
    norm = str(rng.choice(["minmax", "batchnorm", "none"]))
    seed = int(rng.integers(0, 5))
    kinds = [str(k) for k in rng.choice(list(KINDS), int(rng.integers(1, 5)))]
    chs = int(rng.integers(1, 3)); P = int(rng.integers(1, 5))
    n = int(rng.integers(34000, 300000))
    return dict(norm=norm, model_seed=seed, kinds=kinds, chs=chs, P=P, n=n, case=case)


def eval_case(c, rng, dev):
    """-> largest |loss_hip - loss_oracle| over the candidates of one random evaluate() call."""
    om = O.make_synthetic_model(c["model_seed"], input_norm=c["norm"])
    pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, c["norm"])
    pm.load_state_dict(om.state_dict()); pm.eval().to(dev)
    op = O.make_plugins(c["kinds"])
    spec, seen = [], {}
    for k in c["kinds"]:
        seen[k] = seen.get(k, 0) + 1
        spec.append((k if seen[k] == 1 else f"{k}{seen[k]}", KINDS[k][0], KINDS[k][1]))
    pp = E.make_plugins(spec)
    D = sum(p["num_params"] for p in op.values())
    x = O.synth_audio(2000 + c["case"], c["chs"], c["n"])[None]
    tgt = O.synth_audio(3000 + c["case"], c["chs"], c["n"])[None]
    W = rng.random((c["P"], D))
    te_ref = O.get_param_embeds(tgt.clone(), om, SR)
    f_ref, _, _ = O.evaluate(list(W), x, SR, op, te_ref, om)
    te = get_param_embeds(tgt.clone(), pm, SR)
    loss, _, _ = PopulationEvaluator(x, SR, pp, pm, te).evaluate(list(W))
    return float(np.abs(loss.cpu().numpy() - np.array(f_ref)).max())


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=20); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda", 0)
    worst = 0.0
    for case in range(a.cases):
        c = draw_eval_case(rng, case)
        err = eval_case(c, rng, dev)
        worst = max(worst, err)
        print(f"case {case:3d}: {'+'.join(c['kinds']):50s} norm={c['norm']:9s} chs={c['chs']} n={c['n']:6d} P={c['P']}  max |loss diff| {err:.2e}{'' if err < 1e-4 else '   <-- CHECK'}", flush=True)
    print(f"worst loss difference over {a.cases} cases: {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
