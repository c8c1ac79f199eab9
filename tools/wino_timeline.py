#!/usr/bin/env python
"""Phase timeline of one Winograd workgroup (s_memtime stamps recorded by the TRACE instantiation).
    python tools/wino_timeline.py [H W cin cout pool]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
from st_ito import _hip

H, W, cin, cout, pool = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (58, 16, 512, 512, 1)))
S = 512
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
x = torch.randn((S, cin // 8, H, W, 8), device=dev)
w = (torch.randn((cout, cin, 3, 3)) / np.sqrt(9 * cin)).to(dev)
upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, 1), device=dev)
_hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, 1, _hip.ptr(upk), st))
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
out = torch.empty((S, cout // 8, H // 2 if pool else H, W // 2 if pool else W, 8), device=dev)
args = (_hip.ptr(x), _hip.ptr(upk), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), S, H, W, cin, cout, pool, 1, st)
_hip.check(L.stito_conv3x3_bn_relu(*args)); torch.cuda.synchronize()
dbg = torch.zeros(16 * 12 * 8, dtype=torch.int64, device=dev)
_hip.check(L.stito_debug_wino_trace(_hip.ptr(dbg)))
_hip.check(L.stito_conv3x3_bn_relu(*args)); torch.cuda.synchronize()
_hip.check(L.stito_debug_wino_trace(None))
t = dbg.cpu().numpy().reshape(16, 12, 8)
per = np.diff(t[:, 0, 1])
P = np.median(per)
print(f"{H}x{W} {cin}->{cout} pool={pool}: chunk period median {P:.0f}  min {per.min()}  max {per.max()} cycles (MFMA-bound floor: 4096)")
c = t[1:15]
rel = lambda w_, s_: np.median(c[:, w_, s_] - c[:, 0, 1])
for w_ in (0, 1, 4, 7):
    print(f"consumer wave {w_:2d}: at barrier {rel(w_, 0) + P:6.0f} (= prev chunk)  released {rel(w_, 1):5.0f}  mfma done {rel(w_, 2):5.0f}")
for w_ in (8, 9, 10, 11):
    print(f"producer wave {w_:2d}: at barrier {rel(w_, 0) + P:6.0f} (= prev chunk)  released {rel(w_, 1):5.0f}  patch written + U copies issued {rel(w_, 2):5.0f}  "
          f"patch loads issued {rel(w_, 3):5.0f}  transform done {rel(w_, 4):5.0f}")
