#!/usr/bin/env python
"""Depth-first trunk schedule (stito_cnn14_weights.chunk_*, ABI v10): sweep of the chunk size and the run of convs, per-layer ms.

    python tools/chunk_sweep.py [--streams 512] [--frames 469] [--chunks 16,32,64,128] [--runs 2:6,2:5,4:6,...] [--reps 3]

Every configuration runs the whole trunk (stito_cnn14_forward) on the same log-mel-shaped input; conv launches are timed with the
library's own HIP events (stito_conv_timing_read_tagged: the launches of a layer summed over its chunks), the pass with events around
the call; the embeddings are compared bit for bit with the layer-by-layer pass."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from st_ito import _hip
from st_ito.utils import make_synthetic_param_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    ap.add_argument("--frames", type=int, default=469)
    ap.add_argument("--chunks", default="16,32,64,128")
    ap.add_argument("--runs", default="2:6")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--once", default=None, help="chunk:first:last -- two passes of that one schedule and nothing else (for rocprofv3 --pmc runs)")
    a = ap.parse_args()
    L = _hip.lib()
    dev = torch.device("cuda", 0)
    model = make_synthetic_param_model(seed=0, input_norm="minmax")
    g = torch.Generator().manual_seed(3)
    base = torch.rand((8, a.frames, 128), generator=g) * 2 - 1
    lm = torch.stack([base[i % 8] * (0.5 + 0.5 * ((i * 37) % 64) / 64.0) for i in range(a.streams)]).to(dev).contiguous()
    names = ["b1.c1", "b1", "b2.c1", "b2.c2", "b3.c1", "b3.c2", "b4.c1", "b4.c2", "b5.c1", "b5.c2", "b6.c1", "b6.c2"]

    def run(chunk, first, last):
        W, _, _ = model._ensure()   # the schedule is three words of the weights struct: no re-packing
        W.chunk_streams, W.chunk_first_conv, W.chunk_last_conv = chunk, first, last
        for _ in range(2):
            out = model.trunk(lm, a.streams // 2, 2)
        torch.cuda.synchronize()
        per = np.zeros(12)
        tot = 0.0
        cnts = None
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _hip.check(L.stito_conv_timing_enable(1))
            e0.record()
            out = model.trunk(lm, a.streams // 2, 2)
            e1.record()
            torch.cuda.synchronize()
            _hip.check(L.stito_conv_timing_enable(0))
            ms, tag, cnt = (ctypes.c_double * 4096)(), (ctypes.c_int * 4096)(), ctypes.c_int()
            _hip.check(L.stito_conv_timing_read_tagged(ms, tag, 4096, ctypes.byref(cnt)))
            for i in range(cnt.value):
                per[tag[i]] += ms[i]
            cnts = [list(tag[: cnt.value]).count(i) for i in range(12)]
            tot += e0.elapsed_time(e1)
        return per / a.reps, tot / a.reps, [t.clone() for t in out], cnts

    if a.once:
        c, f, l = (int(v) for v in a.once.split(":"))
        W, _, _ = model._ensure()
        W.chunk_streams, W.chunk_first_conv, W.chunk_last_conv = c, f, l
        for _ in range(2):
            model.trunk(lm, a.streams // 2, 2)
        torch.cuda.synchronize()
        return
    print(f"streams {a.streams}, frames {a.frames}, reps {a.reps}; ms per layer (launches of the layer summed), trunk = events around stito_cnn14_forward")
    print(f"{'run':>6} {'chunk':>6} " + " ".join(f"{n:>7}" for n in names[1:]) + f" {'convs':>8} {'trunk':>8}  same bits  launches")
    ref_per, ref_tot, ref_out, cn = run(0, 2, 6)
    print(f"{'-':>6} {'all':>6} " + " ".join(f"{v:7.3f}" for v in ref_per[1:]) + f" {ref_per.sum():8.3f} {ref_tot:8.3f}  {'ref':>9}  {sum(cn)}")
    for r in a.runs.split(","):
        first, last = (int(v) for v in r.split(":"))
        for c in (int(v) for v in a.chunks.split(",")):
            per, tot, out, cn = run(c, first, last)
            same = all(torch.equal(x, y) for x, y in zip(out, ref_out))
            print(f"{r:>6} {c:>6} " + " ".join(f"{v:7.3f}" for v in per[1:]) + f" {per.sum():8.3f} {tot:8.3f}  {str(same):>9}  {sum(cn)}")


if __name__ == "__main__":
    main()
