#!/usr/bin/env python
"""Phase timeline of k_conv_wino8 workgroups (s_memtime stamps recorded by the TRACE instantiation): entry, prologue
done, main loop done, epilogue done and the first barriers, for the workgroups 100, 356, 612, ... of the grid.
    python tools/wino_timeline.py [H W cin cout pool]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
from st_ito import _hip

H, W, cin, cout, pool = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (58, 16, 512, 512, 1)))
algo = int(sys.argv[6]) if len(sys.argv) > 6 else 2   # 1: k_conv_wino8 (F(2x2,3x3)), 2: k_conv_wino43 (F(4x4,3x3))
kch = 8 if algo == 1 else 4
S = 512
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
x = torch.randn((S, cin // 8, H, W, 8), device=dev)
w = (torch.randn((cout, cin, 3, 3)) / np.sqrt(9 * cin)).to(dev)
upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
_hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, algo, _hip.ptr(upk), st))
sc = torch.ones(cout, device=dev); sh = torch.zeros(cout, device=dev)
out = torch.empty((S, cout // 8, H // 2 if pool else H, W // 2 if pool else W, 8), device=dev)
args = (_hip.ptr(x), _hip.ptr(upk), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), S, H, W, cin, cout, pool, algo, st)
_hip.check(L.stito_conv3x3_bn_relu(*args)); torch.cuda.synchronize()
dbg = torch.zeros(8 * 2 * 16, dtype=torch.int64, device=dev)
_hip.check(L.stito_debug_wino_trace(_hip.ptr(dbg)))
_hip.check(L.stito_conv3x3_bn_relu(*args)); torch.cuda.synchronize()
_hip.check(L.stito_debug_wino_trace(None))
t = dbg.cpu().numpy().reshape(8, 2, 16)
n_chunks = cin // kch
floor = 4096 if algo == 1 else 2304
print(f"{H}x{W} {cin}->{cout} pool={pool} algo {algo}: {n_chunks} chunks per workgroup; cycles (s_memtime); MFMA floor per chunk {floor}")
for i in range(8):
    if t[i, 0, 0] == 0:
        continue
    for wv in (0, 1):
        a = t[i, wv]
        per = np.diff(a[4:4 + min(12, n_chunks)])
        print(f"wg {100 + 256 * i:5d} wave {4 * wv}: start {a[0] - t[0, 0, 0]:9d}  prologue {a[1] - a[0]:6d}  loop {a[2] - a[1]:8d} "
              f"({(a[2] - a[1]) / n_chunks:7.1f} / chunk)  epilogue {a[3] - a[2]:6d}  periods {per.tolist()}")
