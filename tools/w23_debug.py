#!/usr/bin/env python
"""Debug aid for conv_wino23r.hip: error of STITO_CONV_WINOGRAD_F2_REG per pixel group (band of 4 output rows x 32 columns)
against float64 torch, with the persistent grid cut to STITO_W23_WG workgroups per channel block.
    STITO_W23_WG=8 python tools/w23_debug.py [n H W cout pool]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
from st_ito import _hip
n, H, W, cout, pool = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (1, 32, 64, 64, 0)
cin, algo = 64, 8
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn((n, cin, H, W), generator=g))
if os.environ.get("W23_DEBUG_CH"):  # keep only the input channels lo:hi (which k-step goes wrong?)
    lo_, hi_ = [int(v) for v in os.environ["W23_DEBUG_CH"].split(":")]
    m_ = torch.zeros(cin); m_[lo_:hi_] = 1.0
    x = x * m_[None, :, None, None]
w = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)
scale = 0.5 + torch.rand(cout, generator=g); shift = 0.2 * torch.randn(cout, generator=g)
ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
if pool: ref = torch.nn.functional.avg_pool2d(ref, 2)
def blocked(t):
    n_, C_, H_, W_ = t.shape
    return t.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()
xd, wd, sd, hd = blocked(x).to(dev), w.contiguous().to(dev), scale.to(dev), shift.to(dev)
packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
_hip.check(L.stito_cnn14_pack_conv(_hip.ptr(wd), cout, cin, algo, _hip.ptr(packed), st))
wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, algo); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.full(blocked(ref).shape, float("nan"), device=dev, dtype=torch.float32)
_hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out), n, H, W, cin, cout, pool, algo, _hip.ptr(ws), wsb, st))
got = out.cpu().double().permute(0, 1, 4, 2, 3).reshape(ref.shape)   # (n, C, Ho, Wo)
err = (got - ref).abs()
err[torch.isnan(got)] = 99.0
rh, cw = (2, 16) if pool else (4, 32)
Ho, Wo = ref.shape[2:]
for s in range(n):
    print(f"stream {s}: max err per group (rows = bands, columns = column blocks); per 32-channel half")
    for b in range((Ho + rh - 1) // rh):
        row = []
        for t in range((Wo + cw - 1) // cw):
            e = err[s, :, b * rh:(b + 1) * rh, t * cw:(t + 1) * cw]
            row.append(" ".join(f"{e[c0:c0 + 32].max().item():8.1e}" for c0 in range(0, cout, 32)))
        print(f"  band {b:3d}: " + " | ".join(row))
