#!/usr/bin/env python
"""Random conv shapes through every algorithm against the direct kernel: stream counts, map heights / widths (ragged tiles,
maps smaller than a tile, blocks straddling stream boundaries), channel counts, pooled or not.  The fixed cases of
tests/test_gpu_parity.py::test_conv_layer_vs_torch are Cnn14's shapes; this looks for geometry corners beside them.
    python tools/conv_fuzz.py [--cases 150] [--seed 0]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
from st_ito import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
rng = np.random.default_rng(a.seed)
bad = ran = 0
for case in range(a.cases):
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 40, 97]))
    H = int(rng.integers(2, 72)); W = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 100, 128]))
    cin = int(rng.choice([8, 16, 24, 64, 128, 192, 264, 512])); cout = int(rng.choice([64, 128, 256, 512, 768, 1024]))
    pool = int(rng.integers(0, 2))
    g = torch.Generator().manual_seed(case)
    x = torch.randn((n, cin // 8, H, W, 8), generator=g).to(dev)
    w = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).to(dev)
    sc = (0.5 + torch.rand(cout, generator=g)).to(dev); sh = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = None
    for m in (0, 1, 2, 3, 4, 5, 8, 9):
        if not L.stito_conv3x3_supported(n, H, W, cin, cout, pool, m):
            continue
        packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, m), device=dev)
        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, m, _hip.ptr(packed), st))
        wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, m)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        out = torch.full((n, cout // 8, Ho, Wo, 8), float("nan"), device=dev)
        _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), n, H, W, cin, cout,
                                              pool, m, _hip.ptr(ws), wsb, st))
        ran += 1
        if m == 0:
            ref = out
            continue
        d = (out - ref).abs()
        tol = 1e-4 * max(1.0, ref.abs().max().item())
        nbad = int((~(d < tol)).sum().item())
        if nbad:
            bad += 1
            print(f"case {case}: n={n} {H}x{W} {cin}->{cout} pool={pool} algo {m}: {nbad} bad elements, max diff {d[~torch.isnan(d)].max().item() if (~torch.isnan(d)).any() else float('nan'):.3e}", flush=True)
print(f"{ran} launches over {a.cases} shapes, mismatching (shape, algo) pairs: {bad}")
