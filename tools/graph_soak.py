#!/usr/bin/env python
"""One fresh process of the hipGraph soak: the fused evaluate step replayed from its captured graph must
return bit-for-bit what the eager launches return, for every replay.

    python tools/graph_soak.py [--pop 32] [--samples 96000] [--replays 20]

Prints one line `graph_soak: OK ...` / `graph_soak: FAIL ...` and exits 0 / 1.  tests/test_gpu_es.py starts many of these."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "st-ito_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pop", type=int, default=32)
    ap.add_argument("--samples", type=int, default=96000)
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds, make_synthetic_param_model

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    pm = make_synthetic_param_model(0)
    g = torch.Generator().manual_seed(1234 + a.seed)
    x = (torch.rand((1, 2, a.samples), generator=g) * 2 - 1) * 0.5
    tgt = (torch.rand((1, 2, a.samples), generator=g) * 2 - 1) * 0.5
    te = get_param_embeds(tgt.clone(), pm, 48000)
    plugins = E.make_plugins("bench5")
    D = 45
    rng = np.random.default_rng(99 + a.seed)
    Ws = [rng.random((a.pop, D)) for _ in range(a.replays)]

    os.environ["STITO_GRAPH"] = "0"
    ev_e = PopulationEvaluator(x, 48000, plugins, pm, te)
    ref = []
    for W in Ws:
        loss, emb, _ = ev_e.evaluate(W)
        ref.append((loss.cpu().numpy().copy(), emb["mid"].cpu().numpy().copy(), emb["side"].cpu().numpy().copy()))
    os.environ["STITO_GRAPH"] = "1"
    ev_g = PopulationEvaluator(x, 48000, plugins, pm, te, capture_after=0)
    bad = []
    for i, W in enumerate(Ws):
        loss, emb, _ = ev_g.evaluate(W)
        got = (loss.cpu().numpy(), emb["mid"].cpu().numpy(), emb["side"].cpu().numpy())
        if not all(np.array_equal(r, q) for r, q in zip(ref[i], got)):
            nb = int((ref[i][0] != got[0]).sum())
            bad.append((i, nb, float(np.abs(ref[i][0] - got[0]).max())))
    if bad:
        print(f"graph_soak: FAIL pop {a.pop} samples {a.samples}: {len(bad)} of {a.replays} replays differ, first {bad[:4]}")
        sys.exit(1)
    print(f"graph_soak: OK pop {a.pop} samples {a.samples}: {a.replays} replays bitwise equal to the eager launches")


if __name__ == "__main__":
    main()
