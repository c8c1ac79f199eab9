#!/usr/bin/env python
"""Time one effect (or a chain) of the render on its own: pop x channels x n samples, HIP events around
stito_render_population.   python tools/fx_bench.py --chain Compressor [--pop 256] [--seconds 10] [--reps 5]
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from st_ito import engine, effects as E

ap = argparse.ArgumentParser()
ap.add_argument("--chain", default="Compressor")
ap.add_argument("--pop", type=int, default=256)
ap.add_argument("--channels", type=int, default=2)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
SR = 48000
dev = torch.device("cuda", 0)
spec = [(f"{k}{i}", getattr(E, "Basic" + k), a.channels) for i, k in enumerate(a.chain.split(","))]
plugins = E.make_plugins(spec, False)
D = sum(p["num_params"] for p in plugins.values())
n = int(a.seconds * SR)
rng = np.random.default_rng(0)
x = torch.from_numpy((0.5 * rng.standard_normal((a.channels, n))).astype(np.float32)).to(dev)
W = torch.from_numpy(rng.random((a.pop, D))).to(dev)
engine.render_population(plugins, x, W, SR)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
for s, e in ev:
    s.record(); engine.render_population(plugins, x, W, SR); e.record()
torch.cuda.synchronize()
ms = [s.elapsed_time(e) for s, e in ev]
print(f"{a.chain}: pop {a.pop} x {a.channels} ch x {n} samples: {np.mean(ms):.3f} ms (min {min(ms):.3f}) incl. the final peak pass")
