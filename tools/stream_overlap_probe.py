#!/usr/bin/env python
"""Do two renders on two HIP streams overlap?  For a chain: time (a) one render of P candidates, (b) two renders of P / 2 one after
the other on one stream, (c) the same two on two streams.  If the chain's kernels were bound by per-launch latency, (c) would
approach half of (b).   python tools/stream_overlap_probe.py [--pop 256]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from st_ito import engine, effects as E

ap = argparse.ArgumentParser()
ap.add_argument("--pop", type=int, default=256)
ap.add_argument("--seconds", type=float, default=10.0)
a = ap.parse_args()
SR, dev = 48000, torch.device("cuda", 0)
n = int(a.seconds * SR)
rng = np.random.default_rng(0)
x = torch.from_numpy((0.5 * rng.standard_normal((2, n))).astype(np.float32)).to(dev)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return min(t)


for chain in ("Compressor", "Reverb", "ParametricEQ", "ParametricEQ,Compressor,Reverb,ParametricEQ,Gain"):
    spec = [(f"{k}{i}", getattr(E, "Basic" + k), 2 if k == "Reverb" else 1) for i, k in enumerate(chain.split(","))]
    plugins = E.make_plugins(spec, False)
    D = sum(p["num_params"] for p in plugins.values())
    W = torch.from_numpy(rng.random((a.pop, D))).to(dev)
    h = a.pop // 2
    outs = [(torch.empty((h, 2, n), device=dev), torch.empty((h,), device=dev)) for _ in range(2)]
    full = (torch.empty((a.pop, 2, n), device=dev), torch.empty((a.pop,), device=dev))
    ch = engine.compile_chain(plugins)

    def one():
        engine.render_population(plugins, x, W, SR, chain=ch, out=full)

    def serial():
        engine.render_population(plugins, x, W[:h], SR, chain=ch, out=outs[0], ws_key="pa")
        engine.render_population(plugins, x, W[h:], SR, chain=ch, out=outs[1], ws_key="pb")

    def overlapped():
        main = torch.cuda.current_stream()
        for s, wsl, o, key in ((s1, W[:h], outs[0], "pa"), (s2, W[h:], outs[1], "pb")):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                engine.render_population(plugins, x, wsl, SR, chain=ch, out=o, ws_key=key)
        main.wait_stream(s1); main.wait_stream(s2)

    print(f"{chain:50s} pop {a.pop}: one launch set {timed(one):7.3f} ms | two halves, one stream {timed(serial):7.3f} ms | two halves, two streams {timed(overlapped):7.3f} ms")
