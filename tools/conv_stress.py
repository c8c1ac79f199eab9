#!/usr/bin/env python
"""Race hunt for the conv kernels: every Cnn14 layer shape at full batch, REPS launches per algorithm, every launch's
output compared with the direct kernel's (outputs pre-filled with NaN so unwritten elements show).
    python tools/conv_stress.py [--streams 512] [--reps 20] [--modes 1,2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import conv_layer_table
from st_ito import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=512)
ap.add_argument("--frames", type=int, default=469)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--modes", default="2,3,4,5,8,9")
ap.add_argument("--layers", default="")
a = ap.parse_args()
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
bad_total = 0
only = {int(v) for v in a.layers.split(",")} if a.layers else None
for li, r in enumerate(r for r in conv_layer_table(a.frames) if r["cin"] % 8 == 0):
    if only is not None and li not in only:
        continue
    g = torch.Generator().manual_seed(li)
    x = torch.randn((a.streams, r["cin"] // 8, r["H"], r["W"], 8), generator=g).to(dev)
    w = (torch.randn((r["cout"], r["cin"], 3, 3), generator=g) / np.sqrt(9 * r["cin"])).to(dev)
    sc = (0.5 + torch.rand(r["cout"], generator=g)).to(dev); sh = (0.1 * torch.randn(r["cout"], generator=g)).to(dev)
    Ho, Wo = (r["H"] // 2, r["W"] // 2) if r["pool"] else (r["H"], r["W"])
    outs = {}
    for m in [0] + [int(v) for v in a.modes.split(",")]:
        if not L.stito_conv3x3_supported(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], m):
            print(f"layer {li}: algo {m} does not cover this shape, skipped", flush=True)
            continue
        packed = torch.empty(L.stito_cnn14_packed_conv_floats(r["cout"], r["cin"], m), device=dev)
        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), r["cout"], r["cin"], m, _hip.ptr(packed), st))
        for rep in range(1 if m == 0 else a.reps):
            out = torch.full((a.streams, r["cout"] // 8, Ho, Wo, 8), float("nan"), device=dev)
            wsb = L.stito_conv3x3_workspace_bytes(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], m)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), a.streams,
                                                  r["H"], r["W"], r["cin"], r["cout"], r["pool"], m, _hip.ptr(ws), wsb, st))
            if m == 0:
                ref = out
                continue
            d = (out - ref).abs()
            nbad = int((~(d < 2e-4 * max(1.0, ref.abs().max().item()))).sum().item())
            if nbad:
                bad_total += 1
                idx = torch.nonzero(~(d < 1e-3))
                print(f"layer {li} {r['H']}x{r['W']} {r['cin']}->{r['cout']} pool={r['pool']} algo {m} rep {rep}: {nbad} bad elements; "
                      f"first {idx[0].tolist()} last {idx[-1].tolist()}; streams hit {sorted(set(idx[:, 0].tolist()))[:8]}", flush=True)
    print(f"layer {li} done", flush=True)
print("BAD LAUNCHES:", bad_total)
