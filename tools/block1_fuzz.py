#!/usr/bin/env python
"""Random shapes through conv_block1 in one launch (stito_conv_block1_f2reg) against the two launches it replaces (k_conv_first,
then k_conv_wino23r on the stored map) and against float64 torch: map sizes 1 .. 200 x 1 .. 200, 1 .. 40 streams with random
per-stream amplitudes (1e-3 .. 1e3) and the occasional all-zero stream, 64 or 128 output channels, pooled or not, every
persistent-grid size the launcher can be asked for.
    python tools/block1_fuzz.py [--cases 200] [--seed 1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
from st_ito import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
rng = np.random.default_rng(a.seed)
F = torch.nn.functional
bad = 0; worst = 0.0; launches = 0
for case in range(a.cases):
    n = int(rng.integers(1, 41)); H = int(rng.integers(1, 201)); W = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 34, 35, 63, 64, 65, 66, 100, 128, 130, 200]))
    cout = int(rng.choice([64, 128])); pool = int(rng.integers(0, 2))
    if pool and (H < 2 or W < 2): pool = 0
    if n * H * W > 3_000_000: n = max(1, 3_000_000 // (H * W))
    c1 = 64
    if not L.stito_conv_block1_f2reg_supported(n, H, W, c1, cout, pool):
        print(f"case {case}: {n}x{H}x{W} -> {cout} pool={pool} not covered", flush=True); continue
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, 1, H, W), generator=g)
    amp = torch.tensor(10.0 ** rng.uniform(-3, 3, n), dtype=torch.float32)
    if n > 2 and rng.random() < 0.3: amp[int(rng.integers(n))] = 0.0
    x *= amp[:, None, None, None]
    w1 = torch.randn((c1, 1, 3, 3), generator=g) / 3.0
    w2 = torch.randn((cout, c1, 3, 3), generator=g) / np.sqrt(9 * c1)
    s1, h1 = 0.5 + torch.rand(c1, generator=g), 0.3 * torch.randn(c1, generator=g)
    s2, h2 = 0.5 + torch.rand(cout, generator=g), 0.2 * torch.randn(cout, generator=g)
    y = torch.relu(F.conv2d(x.double(), w1.double(), padding=1) * s1.double()[None, :, None, None] + h1.double()[None, :, None, None])
    y = torch.relu(F.conv2d(y, w2.double(), padding=1) * s2.double()[None, :, None, None] + h2.double()[None, :, None, None])
    if pool: y = F.avg_pool2d(y, 2)
    n_, C_, H_, W_ = y.shape
    ref = y.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()
    xd = x.reshape(n, H, W).contiguous().to(dev)
    w1d, s1d, h1d, w2d, s2d, h2d = (t.contiguous().to(dev) for t in (w1, s1, h1, w2, s2, h2))
    fw = torch.empty(L.stito_cnn14_packed_conv1_f2reg_floats(), device=dev)
    _hip.check(L.stito_cnn14_pack_conv1_f2reg(_hip.ptr(w1d), _hip.ptr(s1d), _hip.ptr(h1d), c1, _hip.ptr(fw), st))
    upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, c1, 8), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w2d), cout, c1, 8, _hip.ptr(upk), st))
    wsb = L.stito_conv_block1_f2reg_workspace_bytes(n, H, W, c1, cout, pool)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    outs = []
    for wg in (None, "8", "40", "256"):
        if wg: os.environ["STITO_W23_WG"] = wg
        out = torch.full(ref.shape, float("nan"), device=dev)
        _hip.check(L.stito_conv_block1_f2reg(_hip.ptr(xd), _hip.ptr(fw), _hip.ptr(upk), _hip.ptr(s2d), _hip.ptr(h2d), _hip.ptr(out), n, H, W, c1, cout, pool,
                                             _hip.ptr(ws), wsb, st, None))
        os.environ.pop("STITO_W23_WG", None)
        outs.append(out); launches += 1
    got = outs[0].cpu().double()
    ok = not torch.isnan(got).any() and all(torch.equal(o, outs[0]) for o in outs[1:])
    rel = 0.0
    if H_ * W_ > 0:
        for i in range(n):
            m = ref[i].abs().max().item()
            e = (got[i] - ref[i]).abs().max().item()
            rel = max(rel, e / max(m, 1e-30))
    worst = max(worst, rel)
    if not ok or rel > 5e-5:
        bad += 1
        print(f"case {case}: {n}x{H}x{W} 1->64->{cout} pool={pool}: {'grid sizes differ / NaN' if not ok else ''} worst per-stream error {rel:.2e}", flush=True)
print(f"{launches} launches over {a.cases} random shapes: mismatching cases {bad}; worst per-stream error {worst:.2e} of the stream's output maximum (bar 5e-5)")
