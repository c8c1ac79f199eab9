#!/usr/bin/env python
"""End-to-end accuracy of the Cnn14 trunk per conv algorithm (direct / Winograd F(2x2,3x3) / Winograd F(4x4,3x3), all float32 / the
default: F(4x4,3x3) with the layers from 256 output channels up on the f16 matrix pipe with split operands):
embeddings of synthetic stereo audio against the oracle's torch-CPU forward run in float64.
    python tools/trunk_accuracy.py [--seconds 10] [--items 2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import st_ito_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--items", type=int, default=2)
a = ap.parse_args()
n = int(a.seconds * 48000)
x = torch.stack([O.synth_audio(100 + i, 2, n) * (1.0 if i % 2 == 0 else 0.05) for i in range(a.items)])
for norm in ("minmax", "batchnorm"):
    om = O.make_synthetic_model(0, input_norm=norm)
    ref32 = O.get_param_embeds(x.clone(), om, 48000)
    om64 = O.make_synthetic_model(0, input_norm=norm).double()
    ref64 = O.get_param_embeds(x.clone().double(), om64, 48000)
    print(f"input_norm={norm}: oracle float32 vs float64: " + ", ".join(
        f"{k} {((ref32[k].double() - ref64[k]).abs().max() / ref64[k].abs().max()).item():.2e}" for k in ("mid", "side")))
    for algo in ("direct", "winograd", "winograd_f4", "winograd_f4 + split f16 operands (default)"):
        os.environ["STITO_CONV_ALGO"] = algo.split(" ")[0]
        os.environ["STITO_CONV_SPLIT"] = "1" if "split" in algo else "0"
        from st_ito.models.panns import Cnn14
        from st_ito.utils import get_param_embeds
        pm = Cnn14(512, 48000, 2048, 1024, 128, 20, 20000, True, norm)
        pm.load_state_dict(om.state_dict())
        pm.eval().cuda()
        e = get_param_embeds(x.clone(), pm, 48000)
        print(f"  {algo:44s} vs float64 oracle: " + ", ".join(
            f"{k} max rel err {((e[k].double() - ref64[k]).abs().max() / ref64[k].abs().max()).item():.2e}" for k in ("mid", "side"))
            + "   cos distance to ref: " + ", ".join(f"{(1 - torch.cosine_similarity(e[k].double(), ref64[k], dim=-1)).abs().max().item():.1e}" for k in ("mid", "side")))
