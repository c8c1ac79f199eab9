#!/usr/bin/env python
"""bench.py -- candidate-evals/sec of the ES evaluate-population hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., one rank
    per GPU, or from a bare shell: without WORLD_SIZE in the environment bench.py launches those N ranks itself)

One step = one ES iteration on synthetic input: ask() -> render the population through the
effect chain -> log-mel -> AFx-Rep (Cnn14) -> cosine loss -> [all-gather fitness] -> tell().
Workload at N = 1 is BASELINE.json configs[1]: pop = 256, 48 kHz stereo 10 s, 5-effect chain
EQ/comp/reverb/EQ/gain (D = 45), AFx-Rep metric with seeded random weights (the checkpoint is
not available offline).  For N > 1 every GPU evaluates its own 256 candidates of a 256*N
population (weak scaling, configs[3] shape) and the fitness scalars are all-gathered over RCCL.

The JSON line also carries
  roofline     : one object per conv KERNEL FAMILY, never mixed: "f32" = the exact-f32 MFMA layers (peak 157.3 TFLOP/s),
                 "f16-stream" = the split-precision streaming convolutions (f16 hi + lo operands, peak 2 500 TFLOP/s dense,
                 bound by filling LDS: `bound` says so and `lds_fill` carries bytes copied into LDS / time against the L2 -> LDS
                 rate measured with tools/ubench/split_mfma.hip, 24 TB/s, and the guide's L2 figure, 34.5 TB/s),
                 "f16-reg" = the register-resident F(2x2,3x3) layers (same pipe; bound by the instruction issue of ONE wave
                 per SIMD: tools/ubench/w23_shadow.hip).  `roofline` is the family with the largest share of the step, the others sit
                 in `roofline_other`.  achieved = FLOPs of the MFMA instructions those launches actually ISSUE (tile padding and
                 all three split products included; stito_conv3x3_issued_flops) / their summed duration, measured with HIP events
                 the library records on its launch stream around every such launch (stito_conv_timing_enable / _read_each).  The
                 timed steps replay the evaluate step's hipGraph (no host-side launches to bracket), so the events are taken on the
                 same K steps launched eagerly right behind the timed region -- same kernels, same stream, same data shapes;
                 `launch_mode.eager_ms_per_step` is that region's step time.  frac = achieved / peak <= 1.
                 time_at_peak_frac = (time all conv launches would take at the peak of the pipe each runs on) / measured conv time:
                 the one scalar that says how far the conv stack is from its matrix pipes.
                 algorithmic_tflops / end_to_end_algorithmic_frac count the DIRECT-convolution FLOPs 2*9*cin*cout*H*W
                 against the f32 peak: they exceed 1 because Winograd issues 2.25 / 9 (F(4x4,3x3)) or 4 / 9 (F(2x2,3x3)) of those
                 MACs and most layers run on the 16 x faster f16 pipe -- not because work is skipped (the parity tests show it is done).
                 traffic = HBM bytes per launch from the committed rocprofv3 PMC passes, used only if they were taken on THIS
                 kernel source (hash check);
  roofline_dsp : render + log-mel against HBM: SURVEY 8(d)'s algorithmic bytes (8.16 MB per 10 s stereo candidate: shared input
                 read, rendered audio written, log-mel written) / the time of those kernels (HIP events around the two stages
                 of a separate, untimed pass) against 8 TB/s; traffic = FETCH_SIZE x2 + WRITE_SIZE of those kernels per step from the
                 committed PMC passes (profiles/round6_dsp_pmc_traffic.json), quoted only for the sources and workload they were taken on;
  launch_mode  : whether the timed steps replayed the evaluate step's hipGraph, and the step time of the eager region behind them;
  last_fitness_sha16 : sha256 of the last timed step's fitness vector (runs of the same tree are comparable bit for bit);
  stages       : per-rank (evaluate, gather, tell) milliseconds per step, min / max over ranks (diagnosis of a first multi-GPU run);
  cpu_baseline : the CPU oracle (port of the reference path) timed on this box's host cores on a bounded sample of the
                 same workload (rank 0, N = 1 only): mode A = the reference's serial loop (parallel=False), and
                 parallel_pool16 = its mp.Pool(16) render re-created per evaluate call (style_transfer.py:499-502).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

SR = 48000
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (v_mfma_f32_32x32x16_f16), 1024 FLOP/clk/SIMD
PIPE_PEAK = {"f32": MFMA_F32_PEAK_TFLOPS, "f16-stream": MFMA_F16_PEAK_TFLOPS, "f16-reg": MFMA_F16_PEAK_TFLOPS}
FAMILY_BOUND = {"f32": "mfma", "f16-stream": "lds-fill", "f16-reg": "issue"}
HBM_PEAK_TBPS = 8.0           # MI355X_MICROARCH.md (6.29 measured-achievable)
DSP_BYTES_PER_CAND_10S = 8.16e6   # 4 C L (input) + 4 C L (audio) + 4 C T M (log-mel) at L = 480 000


def synth_audio(seed, chs, n):
    """SURVEY.md 8(d) synthetic input (same recipe as the oracle's synth_audio)."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n, dtype=torch.float64) / SR
    tone = 0.2 * torch.sin(2 * np.pi * 110 * t) + 0.1 * torch.sin(2 * np.pi * 440 * t) + 0.05 * torch.sin(2 * np.pi * 3520 * t)
    left = 0.1 * torch.randn(n, generator=g, dtype=torch.float64) + tone
    chans = [left]
    if chs == 2:
        chans.append(0.7 * left + 0.3 * (0.1 * torch.randn(n, generator=g, dtype=torch.float64)))
    x = torch.stack(chans, 0)
    nf = min(32768, n)
    x[..., :nf] = x[..., :nf] * torch.linspace(0, 1, nf, dtype=torch.float64)
    x = x / x.abs().max().clamp(min=1e-8)
    return x.to(torch.float32)


def conv_layer_table(T, M=128):
    """(H, W, cin, cout, pool) of the 12 convs and their algorithmic FLOPs per stream."""
    chans = [1, 64, 128, 256, 512, 1024, 2048]
    H, W, rows = T, M, []
    for b in range(6):
        for j in range(2):
            cin = chans[b] if j == 0 else chans[b + 1]
            cout = chans[b + 1]
            pool = 1 if (j == 1 and b < 5) else 0
            rows.append(dict(H=H, W=W, cin=cin, cout=cout, pool=pool, flops=2.0 * 9 * cin * cout * H * W))
        if b < 5:
            H, W = H // 2, W // 2
    return rows


PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "round6_conv_pmc_traffic.json")


def kernel_source_hash():
    """sha256 over the conv kernels' CODE (// comments and blank lines dropped, so that editing a comment does not orphan a
    measurement): a PMC measurement is only quoted for the source it was taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "st-ito_amd", "csrc")
    for name in ("cnn14.hip", "conv_wino43.hip", "conv_wino23r.hip", "conv_wino23r_body.inc", "conv_wino23r_body_f1.inc", "conv_wino23r_pro.inc", "conv_layout.h", "common.h"):
        for line in open(os.path.join(d, name), "r", encoding="utf-8", errors="replace"):
            code = line.split("//", 1)[0].rstrip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(n_streams):
    """HBM bytes per conv launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of the same 11 launches
    at the same stream count, taken on this bench process: profiles/summarize_pmc_bench.py).  -> (bytes or None, note).
    Refused when the committed file was taken on other sources."""
    try:
        d = json.load(open(PMC_TRAFFIC_JSON))
    except OSError:
        return None, "no committed PMC file"
    if d.get("kernel_source_hash") != kernel_source_hash():
        return None, f"{os.path.basename(PMC_TRAFFIC_JSON)} was taken on kernel sources {d.get('kernel_source_hash')}, this tree is {kernel_source_hash()}"
    if d.get("n_streams") != n_streams:
        return None, f"PMC file is for {d.get('n_streams')} streams"
    return float(d["traffic_bytes_per_launch"]), f"profiles/{os.path.basename(PMC_TRAFFIC_JSON)}"


PMC_DSP_JSON = os.path.join(ROOT, "profiles", "round6_dsp_pmc_traffic.json")


def dsp_source_hash():
    """sha256 over the effect-chain and front-end kernels' code (comments and blank lines dropped), like kernel_source_hash."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "st-ito_amd", "csrc")
    for name in ("dsp.hip", "dsp_view.h", "compressor.hip", "comp_scan.inc", "frontend.hip", "common.h"):
        for line in open(os.path.join(d, name), "r", encoding="utf-8", errors="replace"):
            code = line.split("//", 1)[0].rstrip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def pmc_dsp_traffic(pop, n_samples):
    """HBM bytes per step of the render + log-mel kernels from the committed PMC passes (profiles/summarize_pmc_dsp.py), quoted
    only for the kernel sources and the workload they were taken on.  -> (bytes or None, note)"""
    try:
        d = json.load(open(PMC_DSP_JSON))
    except OSError:
        return None, "no committed PMC file"
    if d.get("dsp_source_hash") != dsp_source_hash():
        return None, f"{os.path.basename(PMC_DSP_JSON)} was taken on sources {d.get('dsp_source_hash')}, this tree is {dsp_source_hash()}"
    if d.get("pop") != pop or d.get("n_samples") != n_samples:
        return None, f"PMC file is for pop {d.get('pop')}, {d.get('n_samples')} samples"
    return float(d["traffic_bytes_per_step"]), f"profiles/{os.path.basename(PMC_DSP_JSON)}"


ALGO_NAMES = {0: "direct", 1: "winograd F(2x2,3x3)", 2: "winograd F(4x4,3x3)", 3: "winograd F(4x4,3x3), input transform hoisted (two kernels)",
              4: "winograd F(4x4,3x3), input transform hoisted, f32 operands as f16 hi + lo on the f16 matrix pipe (3 products, f32 accumulate)",
              5: "the same on 64 x 64 workgroup tiles in two sweeps over the positions",
              9: "the same on 128 x 128 workgroup tiles in six sweeps (one position row each)",
              8: "winograd F(2x2,3x3), weights resident in registers, input transform in registers, split operands on the f16 pipe"}


def conv_layer_times(model, n_streams, T, reps=3):
    """Per-layer table (informational): every conv launch of the trunk timed on its own with HIP events on the launch
    stream (torch's current stream is the stream the C ABI launches on).  -> (layers, per timed launch of one trunk pass
    at n_streams, in launch order: (matrix pipe, MFMA FLOPs issued, bytes copied into LDS by the split-precision kernel))."""
    from st_ito import _hip
    L = _hip.lib()
    W, FE, _ = model._ensure()
    dev = next(model.parameters()).device
    rows = conv_layer_table(T)
    st = _hip.stream_ptr()
    x = torch.randn((n_streams, T, 128), device=dev).clamp_(-1, 1)
    layers = []
    cur = x
    order = []
    fusedr = bool(W.conv1_f2reg_w_dev) and bool(W.conv_wino_dev[1]) and W.conv_wino_algo[1] == 8 and \
        L.stito_conv_block1_f2reg_supported(n_streams, rows[0]["H"], rows[0]["W"], rows[0]["cout"], rows[1]["cout"], 1)
    for i, r in enumerate(rows):
        Ho, Wo = (r["H"] // 2, r["W"] // 2) if r["pool"] else (r["H"], r["W"])
        if fusedr and i == 0:
            continue  # conv_block1 is one launch: reported with its second conv
        out = torch.empty((n_streams, r["cout"] // 8, Ho, Wo, 8), device=dev)
        if fusedr and i == 1:
            wsb = L.stito_conv_block1_f2reg_workspace_bytes(n_streams, r["H"], r["W"], r["cin"], r["cout"], 1)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            args = (_hip.ptr(x), W.conv1_f2reg_w_dev, W.conv_wino_dev[1], W.bn_scale_dev[1], W.bn_shift_dev[1], _hip.ptr(out),
                    n_streams, r["H"], r["W"], r["cin"], r["cout"], 1, _hip.ptr(ws), wsb, st, None)
            _hip.check(L.stito_conv_block1_f2reg(*args))
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in ev:
                a.record()
                _hip.check(L.stito_conv_block1_f2reg(*args))
                b.record()
            torch.cuda.synchronize()
            ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            fl = (r["flops"] + rows[0]["flops"]) * n_streams
            # per pixel group and wave: 96 products of the second conv + 12 of the first (4 x 4 window x 32 channels x 32 pixels each)
            issued = L.stito_conv3x3_issued_flops(n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], 8) * (1.0 + 12.0 / 96.0)
            order.append(("f16-reg", issued, 0.0))
            layers.append(dict(layer="conv_block1 (one launch: conv1 computed on the matrix pipe into conv2's patch ring)", algo=ALGO_NAMES[8],
                               H=r["H"], W=r["W"], cin=1, cout=r["cout"], ms=round(ms, 4), algorithmic_tflops=round(fl / ms / 1e9, 2), pipe="f16-reg",
                               mfma_issued_tflops=round(issued / ms / 1e9, 2), mfma_frac=round(issued / ms / 1e9 / PIPE_PEAK["f16-reg"], 4)))
            cur = out
            continue
        walgo = int(W.conv_wino_algo[i]) if W.conv_wino_algo[i] in (1, 2, 3, 4, 5, 8, 9) else 1
        wino = bool(W.conv_wino_dev[i]) and L.stito_conv3x3_supported(n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], walgo)
        algo = walgo if wino else 0
        wsb = L.stito_conv3x3_workspace_bytes(n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], algo)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        args = (_hip.ptr(cur), W.conv_wino_dev[i] if wino else W.conv_w_dev[i], W.bn_scale_dev[i], W.bn_shift_dev[i],
                _hip.ptr(out), n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], algo, _hip.ptr(ws), wsb, st)
        _hip.check(L.stito_conv3x3_bn_relu_ws(*args))  # warm
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            _hip.check(L.stito_conv3x3_bn_relu_ws(*args))
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        fl = r["flops"] * n_streams
        issued = L.stito_conv3x3_issued_flops(n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], algo)
        pipe = "f16-reg" if algo == 8 else ("f16-stream" if algo in (4, 5, 9) else "f32")
        # split-precision kernel: every operand element of every product travels L2 -> LDS as 4 bytes (hi + lo); per workgroup
        # 36 positions x (32 tiles + 64 couts) x cin elements, i.e. 4 / (2 * 32 * 64 / 96) bytes per f32-equivalent MAC
        # (two-sweep kernel: 128 elements per 64 x 64 MACs)
        # (six-sweep kernel: 256 elements per 128 x 128 MACs)
        per_mac = {4: 96.0 / (32.0 * 64.0), 5: 128.0 / (64.0 * 64.0), 9: 256.0 / (128.0 * 128.0)}
        fill = issued / 3.0 / 2.0 * per_mac[algo] * 4.0 if algo in per_mac else 0.0
        row = dict(layer=f"conv_block{i // 2 + 1}.conv{i % 2 + 1}", algo=ALGO_NAMES[algo] if r["cin"] % 8 == 0 else "direct (VALU, cin = 1)",
                   H=r["H"], W=r["W"], cin=r["cin"], cout=r["cout"], ms=round(ms, 4), algorithmic_tflops=round(fl / ms / 1e9, 2))
        if issued:
            order.append((pipe, issued, fill))
            row["pipe"] = pipe
            row["mfma_issued_tflops"] = round(issued / ms / 1e9, 2)
            row["mfma_frac"] = round(issued / ms / 1e9 / PIPE_PEAK[pipe], 4)
            if fill:
                row["lds_fill_tbps"] = round(fill / ms / 1e9, 2)
        layers.append(row)
        cur = out
    return layers, order


def _pool_render(args):
    """mp.Pool worker of CPU mode B: process_audio of one candidate (style_transfer.py:499-502)."""
    xnp, w, kinds = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import st_ito_oracle as O
    global _POOL_PLUGINS
    try:
        _POOL_PLUGINS
    except NameError:
        _POOL_PLUGINS = {}
    key = tuple(kinds)
    if key not in _POOL_PLUGINS:
        _POOL_PLUGINS[key] = O.make_plugins(list(kinds))
    return O.process_audio(xnp, w, SR, _POOL_PLUGINS[key])


def cpu_baseline(n_samples, kinds, budget_s=12.0):
    """The CPU oracle (a port of the reference path) on a bounded sample of the bench workload.
    Mode A (value): the reference's serial loop, parallel=False (style_transfer.py:504-521): candidates evaluated one
    at a time (C effects + torch-CPU Cnn14) until ~budget_s of CPU work has been spent.
    Mode B (parallel_pool16): parallel=True (499-502): multiprocessing.Pool(16) created and torn down inside every
    evaluate call, the candidates rendered by the pool, one batched Cnn14 forward; 32 candidates per call, 2 calls."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import st_ito_oracle as O
    op = O.make_plugins(kinds)
    om = O.make_synthetic_model(0)
    D = sum(p["num_params"] for p in op.values())
    x = O.synth_audio(1234, 2, n_samples)[None]
    tgt = O.synth_audio(4321, 2, n_samples)[None]
    te = O.get_param_embeds(tgt.clone(), om, SR)
    W = np.random.default_rng(2025).random((64, D))
    O.evaluate([W[0]], x, SR, op, te, om)  # warm (oracle .so build, torch threads)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < len(W) - 1:
        O.evaluate([W[n + 1]], x, SR, op, te, om)
        n += 1
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 4), "unit": "candidate-evals/s", "cores": torch.get_num_threads(),
           "kind": "port", "sample": f"mode A (serial, parallel=False): {n} candidates x 10 s stereo, same chain, oracle.evaluate "
           f"one candidate at a time (C effects + torch-CPU Cnn14, {torch.get_num_threads()} threads of {os.cpu_count()} cpus), {dt:.1f} s"}
    try:
        import multiprocessing as mp
        P_b, calls = 32, 2
        xnp = x[0].numpy()
        t0 = time.perf_counter()
        for c in range(calls):
            with mp.get_context("fork").Pool(processes=16) as pool:   # re-created per evaluate call, like the reference
                audios = pool.map(_pool_render, [(xnp, W[(c * P_b + i) % len(W)], tuple(kinds)) for i in range(P_b)])
            batch = torch.stack([torch.from_numpy(a) for a in audios], 0)
            emb = O.get_param_embeds(batch, om, SR)
            _ = [(-torch.cosine_similarity(emb[k], te[k], dim=-1)) for k in emb]
        dtb = time.perf_counter() - t0
        out["parallel_pool16"] = {"value": round(P_b * calls / dtb, 4), "unit": "candidate-evals/s", "cores": 16,
                                  "sample": f"mode B (parallel=True): {calls} evaluate calls x {P_b} candidates, mp.Pool(16) per call + one "
                                            f"batched torch-CPU Cnn14 forward ({torch.get_num_threads()} threads), {dtb:.1f} s"}
    except Exception as e:  # a box without fork / enough memory: mode A stands alone
        out["parallel_pool16"] = {"value": None, "note": f"not measured: {e}"}
    return out


# The BASELINE.json configurations bench.py can run (the first 8-GPU run must be able to measure the two
# 8-GPU configurations BASELINE names, not only weak-scaled configs[1]).  pop_per_gpu x N = the configuration's population at N = 8.
BASELINE_CONFIGS = {
    1: dict(baseline="1xMI355X, pop=256, 48 kHz stereo 10 s, 5-effect chain (EQ/comp/reverb/EQ/gain), AFx-Rep param metric, 25 iters",
            pop_per_gpu=256, seconds=10.0, steps=5, chain="bench5"),
    3: dict(baseline="8xMI355X, pop=2048 sharded 256/GPU, RCCL all-gather fitness over xGMI, 48 kHz stereo 30 s, 50 iters",
            pop_per_gpu=256, seconds=30.0, steps=50, chain="bench5"),
    4: dict(baseline="8xMI355X, convolution-reverb IR=2 s in chain (partitioned FFT-conv), pop=1024, 48 kHz stereo 30 s -- HBM-bound long-FIR stress",
            pop_per_gpu=128, seconds=30.0, steps=10, chain="bench5-convreverb"),
}


def bench_plugins(chain):
    """-> (plugin dict in the reference's schema, effect kinds for the workload string / the CPU oracle).  "bench5" = EQ / compressor /
    Freeverb / EQ / gain; "bench5-convreverb" = the same with the noise-shaped CONVOLUTION reverb (96 000 taps = 2 s at 48 kHz,
    partitioned FFT convolution: csrc/convreverb.hip) in the reverb's place -- BASELINE.json configs[4]."""
    import functools
    from st_ito import effects as E
    if chain == "bench5":
        return E.make_plugins("bench5"), ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]
    spec = [("ParametricEQ", E.BasicParametricEQ, 1), ("Compressor", E.BasicCompressor, 1),
            ("ConvReverb", functools.partial(E.NoiseShapedReverb, num_samples=96000), 2),
            ("ParametricEQ2", E.BasicParametricEQ, 1), ("Gain", E.BasicGain, 1)]
    return E.make_plugins(spec), ["ParametricEQ", "Compressor", "NoiseShapedReverb(96000 taps)", "ParametricEQ", "Gain"]


def workload_string(args, cfg, world, D, kinds):
    P_total = args.pop_per_gpu * world
    return (f"BASELINE.json configs[{args.config}] ({cfg['baseline']}) as run here: ES evaluate-population, pop={args.pop_per_gpu}/GPU "
            f"({P_total} total), 48 kHz stereo {args.seconds:g} s, chain {'/'.join(kinds)} (D={D}), AFx-Rep Cnn14 (seeded random weights), "
            "CMA-ES seed 42")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 5; --config 3: its 50 iterations; --config 4: 10)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=1, choices=sorted(BASELINE_CONFIGS),
                    help="BASELINE.json configs[] index: 1 = the configuration the metric is quoted on (default); 3 and 4 = the two 8-GPU "
                         "configurations (their per-GPU share at any --gpus)")
    ap.add_argument("--pop-per-gpu", type=int, default=None)
    ap.add_argument("--seconds", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pop512", action="store_true", help="skip the second timed region at BASELINE.json's target population (512 per GPU)")
    args = ap.parse_args()
    cfg = BASELINE_CONFIGS[args.config]
    args.steps = cfg["steps"] if args.steps is None else args.steps
    args.pop_per_gpu = cfg["pop_per_gpu"] if args.pop_per_gpu is None else args.pop_per_gpu
    args.seconds = cfg["seconds"] if args.seconds is None else args.seconds

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started from a bare shell: launch one rank per GPU through torch.distributed.run (what the driver does itself
        # for its scaling runs) and hand its exit code back; rank 0 of the children prints the JSON line
        import subprocess
        # --standalone: the launcher picks a free rendezvous port itself (binding one here and closing it again would leave a
        # window for another process to take it)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or start bench.py without "
                         "WORLD_SIZE in the environment: it launches the ranks itself)")
    if os.environ.get("STITO_BENCH_DRYRUN") == "1":
        # launch-path check without a GPU (tests/test_host_logic.py): rendezvous over gloo, the barrier + max-over-ranks
        # timing skeleton of the real run, one JSON line from rank 0
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo")
            dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        seen = [None] * world
        stages_ms = [10.0 * (rank + 1), 1.0 + rank, 0.5]   # stands in for (evaluate, gather, tell) of the real run
        stages_all = [stages_ms]
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_gather_object(seen, rank)
            stages_all = [None] * world
            dist.all_gather_object(stages_all, stages_ms)
        else:
            seen = [0]
        stages = {name: {"min": min(s_[i] for s_ in stages_all), "max": max(s_[i] for s_ in stages_all)}
                  for i, name in enumerate(("evaluate_ms", "gather_ms", "tell_ms"))}
        if rank == 0:
            pl, kinds_d = bench_plugins(cfg["chain"])
            D_d = sum(p_["num_params"] for p_ in pl.values())
            print(json.dumps({"dryrun": True, "n_gpus": world, "ranks_seen": seen, "max_over_ranks": float(t.item()),
                              "steps": args.steps, "warmup": args.warmup, "stages": stages,
                              "config": {"workload": workload_string(args, cfg, world, D_d, kinds_d), "baseline_config": args.config,
                                         "pop_per_gpu": args.pop_per_gpu, "n_samples": int(round(args.seconds * SR)), "chain": kinds_d}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # STITO_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional check of the N > 1 path on a
    # 1-GPU box; RCCL refuses two ranks on one device).  The driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("STITO_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    # STITO_BENCH_FORCE_DIST=1: at N = 1 build the one-rank process group anyway and send every step through the collective
    # branch (RCCL communicator creation, device binding, all_gather_into_tensor, barrier) -- what a first multi-GPU run needs to work
    force_dist = os.environ.get("STITO_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["STITO_FORCE_COLLECTIVE"] = "1"
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from st_ito import effects as E, _hip
    from st_ito import cmaes
    from st_ito.engine import PopulationEvaluator
    from st_ito.style_transfer import gather_fitness, shard_bounds
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    _hip.lib()

    n = int(round(args.seconds * SR))
    plugins, kinds = bench_plugins(cfg["chain"])
    D = sum(p["num_params"] for p in plugins.values())
    model = make_synthetic_param_model(seed=0, input_norm="minmax")
    x = synth_audio(1234, 2, n)[None]
    # target: the seed-4321 signal rendered through the same chain at w_target = default_rng(7)
    from st_ito.style_transfer import process_audio
    tgt = torch.from_numpy(process_audio(synth_audio(4321, 2, n).numpy(), np.random.default_rng(7).random(D), SR, plugins))[None]
    te = get_param_embeds(tgt, model, SR)
    ev = PopulationEvaluator(x, SR, plugins, model, te, capture_after=0)   # the evaluate step's graph is captured inside the warm-up
    P_total = args.pop_per_gpu * world
    es = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P_total, "seed": 42})

    last_f = []   # fitness vector of the last step (its hash goes into the line: equal runs are comparable bit for bit)
    stage_s = [0.0, 0.0, 0.0]  # evaluate (ask + launches + the sync that ends them), gather, tell -- host clocks of this rank

    def step():
        t_a = time.perf_counter()
        W = es.ask()
        lo, hi = shard_bounds(P_total, rank, world)
        loss, _, _ = ev.evaluate(W[lo:hi])
        es.prefetch()  # the next generation's normal deviates, drawn while the GPU works (as run_es does)
        if world > 1:
            torch.cuda.synchronize()   # N > 1 only: separates this rank's own work from its wait for the slowest rank
        t_b = time.perf_counter()
        f = gather_fitness(loss, P_total).tolist()  # .tolist() = the device->host sync the optimiser needs anyway
        t_c = time.perf_counter()
        last_f[:] = f
        es.tell(W, f)
        t_d = time.perf_counter()
        stage_s[0] += t_b - t_a; stage_s[1] += t_c - t_b; stage_s[2] += t_d - t_c

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timing = rank == 0 and not args.no_roofline
    stage_s[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stages_ms = [1e3 * v / args.steps for v in stage_s]
    import hashlib
    fitness_sha = hashlib.sha256(np.asarray(last_f, dtype=np.float64).tobytes()).hexdigest()[:16]
    conv_each, conv_launches = (ctypes.c_double * 65536)(), ctypes.c_int(0)
    graph_replay = bool(ev._graphs)   # the timed steps replayed the evaluate step's hipGraph (the product default)
    eager_region = not args.no_roofline   # EVERY rank runs it (its steps contain the fitness collective); rank 0 records the events
    if eager_region:
        # The library records HIP events around every MFMA conv launch it makes from the host.  The timed steps above replay the
        # captured graph (no host-side launches), so the per-launch durations come from the SAME steps repeated right behind the
        # timed region with eager launches of the same kernels on the same stream (the CMA-ES simply keeps going); their device
        # time is what rocprofv3 reports for the timed region's launches (profiles/round5_bench_kernel_stats.txt).
        ev_timed = ev if not graph_replay else PopulationEvaluator(x, SR, plugins, model, te, use_graph=False)
        ev_graph, ev = ev, ev_timed
        if graph_replay:
            step()   # warm (eager buffers)
            fence()
        if timing:
            _hip.check(_hip.lib().stito_conv_timing_enable(1))
        t_e = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_eager = time.perf_counter() - t_e
        if timing:
            _hip.check(_hip.lib().stito_conv_timing_enable(0))
            _hip.check(_hip.lib().stito_conv_timing_read_each(conv_each, 65536, ctypes.byref(conv_launches)))
        ev = ev_graph
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # second timed region: BASELINE.json's target line is quoted at pop = 512 (per GPU); configs[1], on which `value` is
    # measured, has 256.  Same chain / input / model, fresh CMA-ES state, same barrier + max-over-ranks bracket.
    pop512 = None
    if not args.no_pop512 and args.pop_per_gpu != 512 and args.config == 1:
        P512 = 512 * world
        es512 = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P512, "seed": 42})

        def step512():
            W = es512.ask()
            lo, hi = shard_bounds(P512, rank, world)
            loss, _, _ = ev.evaluate(W[lo:hi])
            es512.prefetch()
            es512.tell(W, gather_fitness(loss, P512).tolist())
        step512()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step512()
        fence()
        dt512 = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([dt512], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt512 = float(t.item())
        pop512 = {"value": round(P512 * args.steps / dt512, 3), "unit": "candidate-evals/s", "pop_per_gpu": 512,
                  "ms_per_step": round(dt512 / args.steps * 1e3, 3), "steps": args.steps, "warmup": 1}

    ranks_seen = [0]
    stages_all = [stages_ms]
    if dist is not None:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, (rank, torch.cuda.get_device_properties(dev).name, local_dev))
        stages_all = [None] * world
        dist.all_gather_object(stages_all, stages_ms)
    stages = {name: {"min": round(min(s_[i] for s_ in stages_all), 3), "max": round(max(s_[i] for s_ in stages_all), 3)}
              for i, name in enumerate(("evaluate_ms", "gather_ms", "tell_ms"))}
    stages["note"] = ("host clocks per rank and step; evaluate = ask + every launch of the evaluate-population step" +
                      (" + a device sync (N > 1 only, so that gather_ms is the wait for the slowest rank + the all-gather)" if world > 1 else
                       " (asynchronous at N = 1: the device time shows up in gather_ms, whose .tolist() is the step's only sync)"))
    try:
        stages["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if (dist is not None and backend == "nccl") else None
    except Exception:  # noqa: BLE001 -- informational
        stages["rccl_version"] = None

    out = {
        "metric": f"candidate-evals/sec (pop x iters), 48 kHz {args.seconds:g} s stereo, 5-effect chain",
        "value": round(P_total * args.steps / dt, 3), "unit": "candidate-evals/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (effects f32 / f64 as the reference; trunk convs with >= 256 output channels: f32 operands carried as f16 hi + lo pairs "
                 "on the f16 matrix pipe, f32 accumulate -- as close to float64 as the f32 pipe, tools/ubench/split_mfma.hip)",
        "data": "synthetic",
        "config": {"workload": workload_string(args, cfg, world, D, kinds), "baseline_config": args.config,
                   "pop_per_gpu": args.pop_per_gpu, "n_samples": n, "chain": kinds,
                   "parallelism": f"population sharded over {world} GPU(s), fitness all-gather",
                   "backend": backend if dist is not None else None, "ranks": ranks_seen},
        "last_fitness_sha16": fitness_sha,   # of the full fitness vector of the last timed step (every rank holds the same one)
    }
    out["stages"] = stages
    out["launch_mode"] = {"timed_region": "one hipGraph replay per evaluate step (render -> log-mel -> Cnn14 -> loss)" if graph_replay else "eager launches",
                          "eager_ms_per_step": round(dt_eager / args.steps * 1e3, 3) if (timing and graph_replay) else None,
                          "note": "per-launch conv durations (roofline) are HIP events around the eager launches of the same steps, repeated right "
                                  "behind the timed region" if graph_replay else None}
    if pop512 is not None:
        out["north_star_pop512"] = pop512
    if rank == 0:
        if not args.no_roofline:
            T = n // 1024 + 1
            streams_per_launch = min(2 * args.pop_per_gpu, model.max_streams_per_pass)
            passes_per_step = (2 * args.pop_per_gpu + streams_per_launch - 1) // streams_per_launch
            layers, order = conv_layer_times(model, streams_per_launch, T)
            # algorithmic conv FLOPs (direct-convolution count 2*9*cin*cout*H*W) of the MFMA launches of one step on this rank
            fl_step = sum(r["flops"] for r in conv_layer_table(T) if r["cin"] % 8 == 0) * 2 * args.pop_per_gpu
            n_timed = min(conv_launches.value, 65536)
            # the timed launches repeat the pass's launch order; the last pass of a step may be partial (fewer streams): its
            # issued work is scaled by its share of the streams
            per_pass = len(order)
            fam = {p: dict(ms=0.0, flops=0.0, fill=0.0, n=0) for p in PIPE_PEAK}
            n_full, rem = divmod(2 * args.pop_per_gpu, streams_per_launch)
            pass_scale = [1.0] * n_full + ([rem / streams_per_launch] if rem else [])
            conv_ms_total = 0.0
            for i in range(n_timed):
                pipe, issued, fill = order[i % per_pass]
                sc = pass_scale[(i // per_pass) % passes_per_step]
                f = fam[pipe]
                f["ms"] += conv_each[i]; f["flops"] += issued * sc; f["fill"] += fill * sc; f["n"] += 1
                conv_ms_total += conv_each[i]
            traffic, traffic_note = pmc_traffic_per_launch(streams_per_launch) if (args.config == 1 and n == 480000) else (None, "the committed PMC passes are of configs[1]")

            def family(pipe):
                f = fam[pipe]
                if not f["n"]:
                    return None
                ach = f["flops"] / f["ms"] / 1e9
                names = sorted({l["algo"] for l in layers if l.get("pipe") == pipe})
                d = {"bound": FAMILY_BOUND[pipe], "pipe": pipe,
                     "kernel": f"the {f['n'] // max(args.steps * passes_per_step, 1)} 3x3-conv launches of a trunk pass in this family (" + "; ".join(names) +
                               "). achieved = FLOPs of the MFMA instructions actually issued (tile padding included) / launch time",
                     "achieved": round(ach, 2), "peak": PIPE_PEAK[pipe], "unit": "TFLOP/s", "frac": round(ach / PIPE_PEAK[pipe], 4),
                     "flops_per_launch": f["flops"] / f["n"], "avg_launch_ms": round(f["ms"] / f["n"], 4), "launches_timed": f["n"],
                     "share_of_step": round(f["ms"] / (dt * 1e3), 4)}
                if pipe == "f16-stream":
                    d["bound_note"] = ("bound by filling LDS from L2, not by the matrix pipe: 4 bytes per operand element (f16 hi + lo) for 21.3 "
                                       "(one sweep) / 32 (two sweeps) MACs; lds_fill = those bytes / launch time (the transform pass is inside the "
                                       "time).  Ceilings: 24 TB/s = the L2 -> LDS rate this build measured for L2-resident streams "
                                       "(tools/ubench/split_mfma.hip, profiles/round3_split_mfma_ubench.txt); 34.5 TB/s = the guide's L2 bandwidth.  "
                                       "What the L2 misses are (memory-side counters, profiles/round5_conv_ea_pmc.txt): 2.1 - 2.9 TB/s of reads at "
                                       "830 - 1 230 L2 clocks, a 30 - 95 % HBM / Infinity-Cache mix far below the fabric's 6.5 - 7.6 TB/s: they cost the "
                                       "CU's copy queue latency slots, not bandwidth, so the bound stays the L2 -> LDS copy rate")
                    fill_tbps = f["fill"] / f["ms"] / 1e9
                    d["lds_fill"] = {"achieved": round(fill_tbps, 2), "peak": 24.0, "peak_source": "self-measured (split_mfma ubench)", "unit": "TB/s",
                                     "frac": round(fill_tbps / 24.0, 4), "frac_of_guide_l2_34.5": round(fill_tbps / 34.5, 4)}
                if pipe == "f16-reg":
                    d["bound_note"] = ("one wave per SIMD (512 registers: the transformed weights live in them) issues one instruction per ~8 cycles "
                                       "whatever its type (tools/ubench/w23_shadow.hip): ~900 instructions per 96 products of 32 cycles each")
                return d

            fams = {p_: family(p_) for p_ in PIPE_PEAK}
            ranked = sorted((p_ for p_ in fams if fams[p_] is not None), key=lambda p_: -fam[p_]["ms"])
            at_peak_ms = sum(fam[p_]["flops"] / (PIPE_PEAK[p_] * 1e9) for p_ in ranked)
            out["roofline"] = dict(fams[ranked[0]])
            out["roofline"].update({
                "time_at_peak_frac": round(at_peak_ms / conv_ms_total, 4),
                "time_at_peak_note": "time all MFMA conv launches would take at the dense peak of the pipe each runs on / their measured time",
                "algorithmic_tflops": round(fl_step * args.steps / conv_ms_total / 1e9, 2),
                "algorithmic_note": "direct-convolution FLOPs (2*9*cin*cout*H*W) of ALL MFMA conv launches / their time; above any peak "
                                    "because Winograd issues 2.25 / 9 or 4 / 9 of those MACs and most layers run on the 16 x faster f16 pipe",
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: FETCH_SIZE x2 + WRITE_SIZE), all conv launches", "traffic_source": traffic_note,
                "n_streams": streams_per_launch,
                "conv_share_of_step": round(conv_ms_total / (dt * 1e3), 4),
                "conv_ms_per_step": round(conv_ms_total / args.steps, 3),
                # whole path (DSP + front end + trunk + host) in direct-convolution FLOPs against the f32 peak
                "end_to_end_algorithmic_frac": round(fl_step / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "end_to_end_note": "> 1 is Winograd + the pipe change (see algorithmic_note), not skipped work",
                "layers": layers,
            })
            if len(ranked) > 1:
                out["roofline_other"] = [fams[p_] for p_ in ranked[1:]]
            # ---- render + log-mel against HBM (a separate, untimed pass with events around the two stages) ----
            from st_ito.engine import render_population
            Wd = torch.rand((args.pop_per_gpu, D), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(2025))
            xin = ev._input(False, np.random)[0]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            audio = peaks = None
            for it in range(2):   # the second pass is the measured one
                evs[0].record()
                audio, peaks = render_population(plugins, xin, Wd, SR, chain=ev.chain)
                evs[1].record()
                step_c = max(1, model.max_streams_per_pass // audio.shape[1])
                for b0 in range(0, audio.shape[0], step_c):
                    model.logmel(audio[b0:b0 + step_c], peaks[b0:b0 + step_c].contiguous(), 2)
                evs[2].record()
                torch.cuda.synchronize()
            r_ms, l_ms = evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2])
            dsp_bytes = DSP_BYTES_PER_CAND_10S * (n / 480000.0) * args.pop_per_gpu
            dsp_traffic, dsp_traffic_note = pmc_dsp_traffic(args.pop_per_gpu, n)
            out["roofline_dsp"] = {"bound": "hbm", "kernel": "effect-chain render (k_eq, compressor, k_reverb, k_eq + gain + peak) + k_logmel_wave",
                                   "achieved": round(dsp_bytes / ((r_ms + l_ms) * 1e-3) / 1e12, 3), "peak": HBM_PEAK_TBPS, "unit": "TB/s",
                                   "frac": round(dsp_bytes / ((r_ms + l_ms) * 1e-3) / 1e12 / HBM_PEAK_TBPS, 4),
                                   "algorithmic_bytes": dsp_bytes, "render_ms": round(r_ms, 3), "logmel_ms": round(l_ms, 3),
                                   "share_of_step": round((r_ms + l_ms) / (dt / args.steps * 1e3), 4),
                                   "traffic": dsp_traffic, "traffic_unit": "HBM bytes per step (PMC: FETCH_SIZE x2 + WRITE_SIZE), render + log-mel kernels",
                                   "traffic_source": dsp_traffic_note,
                                   "note": "algorithmic bytes of SURVEY 8(d) (input read once, audio written once, log-mel written) / time; the "
                                           "time-serial effects (float64 biquad cascade, envelope follower, comb / all-pass lines) are latency- "
                                           "and issue-bound as SURVEY 8(d) expected: DESIGN.md 4.2 gives the per-kernel account"}
        if world == 1 and not args.no_cpu_baseline and args.config == 1:
            out["cpu_baseline"] = cpu_baseline(n, kinds)
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = {"value": None, "note": "the CPU port is timed on configs[1], the configuration the metric is quoted on (run without --config)"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
