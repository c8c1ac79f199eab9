#!/usr/bin/env python
"""conv_block1 at the bench shape (512 streams x 469 x 128, 1 -> 64 -> 64, pooled): the one-launch form on the register-resident
F(2x2,3x3) kernel (stito_conv_block1_f2reg) against its two launches (k_conv_first, then the same kernel on the stored map).
    python tools/block1_bench.py [--streams 512] [--frames 469] [--reps 5]
STITO_W23_CLK=1 (+ a -DW23_TRACE=1 build through STITO_LIB_PATH) prints the per-phase timeline of the kernels."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import numpy as np, torch
os.environ.setdefault("STITO_W23_AMAX_ONCE", "1")
from st_ito import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=512)
ap.add_argument("--frames", type=int, default=469)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
L = _hip.lib(); dev = torch.device("cuda", 0); st = _hip.stream_ptr()
n, H, W, c1, cout, pool = a.streams, a.frames, 128, 64, 64, 1
g = torch.Generator().manual_seed(0)
x = torch.randn((n, H, W), generator=g).clamp_(-1, 1).to(dev)
w1 = (torch.randn((c1, 1, 3, 3), generator=g) / 3.0).to(dev)
w2 = (torch.randn((cout, c1, 3, 3), generator=g) / np.sqrt(9 * c1)).to(dev)
s1, h1 = (0.5 + torch.rand(c1, generator=g)).to(dev), (0.3 * torch.randn(c1, generator=g)).to(dev)
s2, h2 = (0.5 + torch.rand(cout, generator=g)).to(dev), (0.2 * torch.randn(cout, generator=g)).to(dev)
fw = torch.empty(L.stito_cnn14_packed_conv1_f2reg_floats(), device=dev)
_hip.check(L.stito_cnn14_pack_conv1_f2reg(_hip.ptr(w1), _hip.ptr(s1), _hip.ptr(h1), c1, _hip.ptr(fw), st))
upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, c1, 8), device=dev)
_hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w2), cout, c1, 8, _hip.ptr(upk), st))
pk1 = torch.empty(L.stito_cnn14_packed_conv_floats(c1, 1, 0), device=dev)
_hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w1), c1, 1, 0, _hip.ptr(pk1), st))
wsb = L.stito_conv_block1_f2reg_workspace_bytes(n, H, W, c1, cout, pool)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
out = torch.empty((n, cout // 8, H // 2, W // 2, 8), device=dev)
out2 = torch.empty_like(out)
mid = torch.empty((n, c1 // 8, H, W, 8), device=dev)
amax = torch.zeros(n, dtype=torch.int32, device=dev)
wsb2 = L.stito_conv3x3_workspace_bytes(n, H, W, c1, cout, pool, 8)
ws2 = torch.empty(max(wsb2, 16), dtype=torch.uint8, device=dev)


def fused():
    _hip.check(L.stito_conv_block1_f2reg(_hip.ptr(x), _hip.ptr(fw), _hip.ptr(upk), _hip.ptr(s2), _hip.ptr(h2), _hip.ptr(out), n, H, W, c1, cout, pool,
                                         _hip.ptr(ws), wsb, st, None))


def first():
    _hip.check(L.stito_conv3x3_bn_relu(_hip.ptr(x), _hip.ptr(pk1), _hip.ptr(s1), _hip.ptr(h1), _hip.ptr(mid), n, H, W, 1, c1, 0, 0, st))


def second():   # STITO_W23_AMAX_ONCE=1: the stream maxima of the first call are kept (inside the trunk the first conv reports them)
    _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(mid), _hip.ptr(upk), _hip.ptr(s2), _hip.ptr(h2), _hip.ptr(out2), n, H, W, c1, cout, pool, 8,
                                          _hip.ptr(ws2), wsb2, st))


def timeit(f):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for e0, e1 in ev:
        e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))


os.environ.setdefault("STITO_W23_AMAX_ONCE", "1")
have_full = True
t_f = timeit(fused)
print(f"one launch (stito_conv_block1_f2reg): {t_f:.3f} ms")
if have_full:
    t1, t2 = timeit(first), timeit(second)
    print(f"two launches: k_conv_first {t1:.3f} ms + k_conv_wino23r {t2:.3f} ms = {t1 + t2:.3f} ms")
    d = (out - out2).abs().max().item()
    print(f"max |one launch - two launches| = {d:.3e} (output max {out2.abs().max().item():.3f})")
