#!/usr/bin/env python
"""Per-layer timing of the Cnn14 conv stack for the available conv kernel variants.
    python tools/conv_bench.py [--streams 128] [--modes 0,1,2] [--frames 469]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import conv_layer_table
from st_ito import _hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=128)
    ap.add_argument("--frames", type=int, default=469)
    ap.add_argument("--modes", default="0,1", help="conv algorithms: 0 direct, 1 winograd F(2x2,3x3), 2 winograd F(4x4,3x3), 3 = 2 with the input transform hoisted, 4 = 3 on the f16 pipe with split operands, 5 = 4 on 64 x 64 tiles in two sweeps, 8 = F(2x2,3x3) with register-resident weights (64 input channels; production mix elsewhere), 9 = the same on 128 x 128 tiles in six sweeps (cout % 512 == 0; production mix elsewhere), 99 = round 3's mix (cout >= 256: 5 from cin 512 up, else 4; below: 2), 100 = the production mix (8 on 64 input channels, 9 from 512 input channels, else as 99)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--max-cin", type=int, default=1 << 30, help="only the layers with at most this many input channels")
    a = ap.parse_args()
    L = _hip.lib()
    dev = torch.device("cuda", 0)
    st = _hip.stream_ptr()
    rows = [r for r in conv_layer_table(a.frames) if r["cin"] % 8 == 0 and r["cin"] <= a.max_cin]
    # mode 100 = the production mix of st_ito/models/panns.py: 8 on the 64-input-channel layers, 9 from 512 input channels where cout % 512 == 0,
    # else 5 from 512 input channels, 4 from 256 output channels, 2 below; mode 99 = the streaming kernels of round 3 (5 / 4 / 2) -- what the
    # modes 8 and 9 fall back to where they do not cover a shape, so that either is measured against the kernels it replaced
    modes = [int(m) for m in a.modes.split(",")]
    def sup(r, m): return L.stito_conv3x3_supported(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], m)
    def algo_of(m, r):
        old_mix = (5 if r["cin"] >= 512 else 4) if r["cout"] >= 256 else 2
        if m == 100:
            m = 8 if sup(r, 8) else (9 if r["cin"] >= 512 and sup(r, 9) else old_mix)
        if m == 99 or (m in (8, 9) and not sup(r, m)):
            m = old_mix
        if m in (4, 5) and not L.stito_conv3x3_supported(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], m):
            m = 3
        if m == 3 and not L.stito_conv3x3_supported(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], 3):
            return 2  # the hoisted transform needs cout % 256 == 0
        return m
    res = {m: [] for m in modes}
    ref_out = {}
    for li, r in enumerate(rows):
        g = torch.Generator(device="cpu").manual_seed(li)
        x = torch.randn((a.streams, r["cin"] // 8, r["H"], r["W"], 8), generator=g).to(dev)
        w = (torch.randn((r["cout"], r["cin"], 3, 3), generator=g) / np.sqrt(9 * r["cin"])).to(dev)
        sc = (0.5 + torch.rand(r["cout"], generator=g)).to(dev)
        sh = (0.1 * torch.randn(r["cout"], generator=g)).to(dev)
        Ho, Wo = (r["H"] // 2, r["W"] // 2) if r["pool"] else (r["H"], r["W"])
        for mode in modes:
            m = algo_of(mode, r)
            packed = torch.empty(L.stito_cnn14_packed_conv_floats(r["cout"], r["cin"], m), device=dev)
            _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), r["cout"], r["cin"], m, _hip.ptr(packed), st))
            out = torch.empty((a.streams, r["cout"] // 8, Ho, Wo, 8), device=dev)
            wsb = L.stito_conv3x3_workspace_bytes(a.streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], m)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            args = (_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), a.streams, r["H"], r["W"],
                    r["cin"], r["cout"], r["pool"], m, _hip.ptr(ws), wsb, st)
            _hip.check(L.stito_conv3x3_bn_relu_ws(*args))
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for s, e in ev:
                s.record(); _hip.check(L.stito_conv3x3_bn_relu_ws(*args)); e.record()
            torch.cuda.synchronize()
            ms = float(np.median([s.elapsed_time(e) for s, e in ev]))
            res[mode].append((ms, r["flops"] * a.streams / ms / 1e9))
            if mode == modes[0]:
                ref_out[li] = out.clone()
            else:
                err = (out - ref_out[li]).abs().max().item()
                assert err < 1e-4 * max(1.0, ref_out[li].abs().max().item()), f"layer {li}: mode {mode} differs from mode {modes[0]} by {err}"
    print(f"{'layer':28s}" + "".join(f"  algo{m}: ms   TF/s" for m in modes))
    for li, r in enumerate(rows):
        name = f"{r['H']}x{r['W']} {r['cin']}->{r['cout']}{' pool' if r['pool'] else ''}"
        print(f"{name:28s}" + "".join(f"  {res[m][li][0]:9.3f} {res[m][li][1]:6.1f}" for m in modes))
    tot_fl = sum(r["flops"] for r in rows) * a.streams
    for m in modes:
        t = sum(x[0] for x in res[m])
        print(f"algo {m}: total {t:.2f} ms  {tot_fl / t / 1e9:.1f} TFLOP/s  ({tot_fl / t / 1e9 / 157.3 * 100:.1f}% of f32 MFMA peak)")


if __name__ == "__main__":
    main()
