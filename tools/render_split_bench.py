#!/usr/bin/env python
"""Is the render of a population faster as two half populations on two HIP streams (their kernels are latency-bound and
leave issue slots and CUs idle) than as one call?   python tools/render_split_bench.py [--pop 256] [--reps 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from st_ito import engine, effects as E

ap = argparse.ArgumentParser()
ap.add_argument("--pop", type=int, default=256)
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
SR = 48000
dev = torch.device("cuda", 0)
plugins = E.make_plugins("bench5")
chain = engine.compile_chain(plugins, False)
D = chain[1]
n = int(a.seconds * SR)
rng = np.random.default_rng(0)
x = torch.from_numpy((0.5 * rng.standard_normal((2, n))).astype(np.float32)).to(dev)
W = torch.from_numpy(rng.random((a.pop, D))).to(dev)
ref_audio, ref_peaks = engine.render_population(plugins, x, W, SR, chain=chain)
audio = torch.empty_like(ref_audio); peaks = torch.empty_like(ref_peaks)
streams = [torch.cuda.Stream(dev) for _ in range(a.parts)]
main = torch.cuda.current_stream(dev)
bounds = [(i * a.pop // a.parts, (i + 1) * a.pop // a.parts) for i in range(a.parts)]

def split():
    for s in streams:
        s.wait_stream(main)
    for i, (p0, p1) in enumerate(bounds):
        with torch.cuda.stream(streams[i]):
            engine.render_population(plugins, x, W[p0:p1], SR, chain=chain, out=(audio[p0:p1], peaks[p0:p1]), ws_key=f"render{i}")
    for s in streams:
        main.wait_stream(s)

def whole():
    engine.render_population(plugins, x, W, SR, chain=chain, out=(audio, peaks))

for name, fn in (("one call", whole), (f"{a.parts} parts on {a.parts} streams", split), ("one call", whole), (f"{a.parts} parts on {a.parts} streams", split)):
    fn(); torch.cuda.synchronize()
    ok = torch.equal(audio, ref_audio) and torch.equal(peaks, ref_peaks)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in ev]
    print(f"{name:28s}: {np.mean(ms):7.3f} ms (min {min(ms):.3f})  bitwise equal to the single call: {ok}")
