#!/usr/bin/env python
"""Timing of the log-mel front end alone at the bench shape (256 candidates x stereo x 10 s at 48 kHz):
the wave-per-frame kernel (k_logmel_wave) against the general kernel (STITO_LOGMEL_GENERIC=1).
    python tools/logmel_bench.py [--pop 256] [--reps 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))
import torch
from st_ito.utils import make_synthetic_param_model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = make_synthetic_param_model(seed=0, input_norm="minmax").to(dev)
    x = torch.randn(a.pop, 2, 480000, device=dev) * 0.1
    for gen in ("1", "0"):
        os.environ["STITO_LOGMEL_GENERIC"] = gen
        model.logmel(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            model.logmel(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print(f"{'general kernel' if gen == '1' else 'wave kernel   '}: {ms:.3f} ms  ({x.numel() * 4 / ms / 1e9:.2f} TB/s of audio read once)")


if __name__ == "__main__":
    main()
