#!/usr/bin/env python
"""Re-run one case of tools/soak.py and report the error after every prefix of its chain (is a large final error
conditioning -- tanh(x * 10^(48/20)) multiplies differences by up to 251 per distortion -- or a defect of one stage?).
    python tools/soak_case.py --seed 2 --case 45"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import st_ito_oracle as O
from st_ito import effects as E, engine
from soak import KINDS, SR   # (replays soak.py's generator with --min-fx 1)

ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=0); ap.add_argument("--case", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
dev = torch.device("cuda", 0)
for case in range(a.case + 1):   # replay the generator exactly like soak.py
    n_fx = int(rng.integers(1, 6)); kinds = [str(k) for k in rng.choice(list(KINDS), n_fx)]
    with_bypass = bool(rng.integers(0, 2)); ns = bool(rng.integers(0, 2)); chs = int(rng.integers(1, 3))
    n = int(rng.choice([1, 5, 191, 193, 4096, 4097, 30011, 48000, 65536, 100003])); P = int(rng.integers(1, 5))
    fixed = rng.integers(0, 3) == 0 and "Compressor" in kinds
    D = sum(len(KINDS[k][0]().parameters) + int(with_bypass) for k in kinds)
    scale = float(rng.uniform(0.05, 1.0)); W = rng.random((P, D))
print(kinds, "chs", chs, "n", n, "P", P, "bypass", with_bypass, "ns", ns)
x = (O.synth_audio(1000 + a.case, chs, max(n, 2))[:, :n] * scale).numpy()
for m in range(1, len(kinds) + 1):
    sub = kinds[:m]
    op = O.make_plugins(sub, with_bypass)
    spec, seen = [], {}
    for k in sub:
        seen[k] = seen.get(k, 0) + 1
        spec.append((k if seen[k] == 1 else f"{k}{seen[k]}", KINDS[k][0], KINDS[k][1]))
    pp = E.make_plugins(spec, with_bypass)
    Dm = sum(p["num_params"] for p in op.values())
    errs = []
    for p in range(P):
        w = W[p][:Dm]
        ref = O.process_audio(x.copy(), w, SR, op, normalize_stages=ns)
        audio, peaks = engine.render_population(pp, torch.from_numpy(x).to(dev), torch.from_numpy(w[None]).to(dev), SR, chain=engine.compile_chain(pp, ns))
        engine.normalize_audio_(audio, peaks)
        errs.append(float(np.abs(audio[0].cpu().numpy() - ref).max()))
    print(f"prefix {m} ({'+'.join(sub)}): max err per candidate {['%.2e' % e for e in errs]}")
