#!/usr/bin/env python
"""Timing of the convolution-reverb stage at the BASELINE.json configs[4] per-GPU shape
(pop 128 = 1024 / 8 GPUs, 48 kHz stereo 30 s, IR 96 000 taps), as first effect (shared input
spectra) and after an in-place effect (per-candidate input spectra).
    python tools/convreverb_bench.py [--pop 128] [--seconds 30] [--taps 96000]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import synth_audio
from st_ito import effects as E, engine


def plug(name, inst, nch):
    names = list(inst.parameters.keys())
    return {name: {"class_path": type(inst), "num_params": len(names), "num_channels": nch, "fixed_parameters": {},
                   "instance": inst, "parameter_names": names}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--taps", type=int, default=96000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = int(a.seconds * 48000)
    x = synth_audio(1, 2, n).to(dev)
    rv = E.NoiseShapedReverb(num_samples=a.taps)
    first = plug("ConvReverb", rv, 2)
    after = {**plug("Gain", E.BasicGain(), 1), **plug("ConvReverb", rv, 2)}
    for label, pl in (("first in chain (shared input spectra)", first), ("after gain (per-candidate spectra)", after)):
        D = sum(p["num_params"] for p in pl.values())
        W = torch.from_numpy(np.random.default_rng(0).random((a.pop, D))).to(dev)
        engine.render_population(pl, x, W, 48000)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(3):
            ev[i].record()
            engine.render_population(pl, x, W, 48000)
        ev[3].record()
        torch.cuda.synchronize()
        ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
        J, K = (n + 4095) // 4096, (a.taps + 4095) // 4096
        mac = a.pop * 2 * J * K * 4096 * 8 / 1e9
        print(f"{label}: {ms:8.2f} ms  ({a.pop} cand x {a.seconds:g} s stereo, {a.taps} taps: J={J} K={K}, "
              f"spectral MAC {mac:.0f} GFLOP -> {mac / ms:.1f} TFLOP/s, audio {a.pop * 2 * n * 8 / 1e9:.2f} GB r+w)")


if __name__ == "__main__":
    main()
