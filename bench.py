#!/usr/bin/env python
"""bench.py -- candidate-evals/sec of the ES evaluate-population hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one ES iteration on synthetic input: ask() -> render the population through the
effect chain -> log-mel -> AFx-Rep (Cnn14) -> cosine loss -> [all-gather fitness] -> tell().
Workload at N = 1 is BASELINE.json configs[1]: pop = 256, 48 kHz stereo 10 s, 5-effect chain
EQ/comp/reverb/EQ/gain (D = 45), AFx-Rep metric with seeded random weights (the checkpoint is
not available offline).  For N > 1 every GPU evaluates its own 256 candidates of a 256*N
population (weak scaling, configs[3] shape) and the fitness scalars are all-gathered over RCCL.

The JSON line also carries
  roofline     : the f32-MFMA conv kernel family (k_conv_wino / k_conv3x3): algorithmic FLOPs per
                 launch / average launch duration, measured with HIP events recorded by the library
                 on its launch stream around every conv launch of the TIMED steps
                 (stito_conv_timing_enable/read), against the 157.3 TFLOP/s f32-MFMA peak; traffic =
                 HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/);
  cpu_baseline : the CPU oracle (port of the reference path) timed on this box's host cores on
                 a bounded sample of the same workload (rank 0, N = 1 only).
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


PMC_TRAFFIC_JSON = os.path.join(ROOT, "profiles", "round1_p_conv_pmc_traffic.json")


def pmc_traffic_per_launch(n_streams):
    """HBM bytes per conv launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of the
    same 11 launches at the same stream count), or None if no committed measurement matches."""
    try:
        d = json.load(open(PMC_TRAFFIC_JSON))
    except OSError:
        return None
    return float(d["traffic_bytes_per_launch"]) if d.get("n_streams") == n_streams else None


def conv_layer_times(model, n_streams, T, reps=3):
    """Per-layer table (informational): every conv launch of the trunk timed on its own with HIP
    events on the launch stream (torch's current stream is the stream the C ABI launches on)."""
    from st_ito import _hip
    L = _hip.lib()
    W, FE, _ = model._ensure()
    dev = next(model.parameters()).device
    rows = conv_layer_table(T)
    st = _hip.stream_ptr()
    x = torch.randn((n_streams, T, 128), device=dev).clamp_(-1, 1)
    layers = []
    cur = x
    tot_flops = tot_ms = 0.0
    mfma_flops = mfma_ms = 0.0
    for i, r in enumerate(rows):
        Ho, Wo = (r["H"] // 2, r["W"] // 2) if r["pool"] else (r["H"], r["W"])
        out = torch.empty((n_streams, r["cout"] // 8, Ho, Wo, 8), device=dev)
        wino = bool(W.conv_wino_dev[i]) and L.stito_conv3x3_supported(n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], 1)
        args = (_hip.ptr(cur), W.conv_wino_dev[i] if wino else W.conv_w_dev[i], W.bn_scale_dev[i], W.bn_shift_dev[i],
                _hip.ptr(out), n_streams, r["H"], r["W"], r["cin"], r["cout"], r["pool"], 1 if wino else 0, st)
        _hip.check(L.stito_conv3x3_bn_relu(*args))  # warm
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            _hip.check(L.stito_conv3x3_bn_relu(*args))
            b.record()
        torch.cuda.synchronize()
        ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        fl = r["flops"] * n_streams
        layers.append(dict(layer=f"conv_block{i // 2 + 1}.conv{i % 2 + 1}", algo="winograd" if wino else "direct", H=r["H"], W=r["W"], cin=r["cin"], cout=r["cout"],
                           ms=round(ms, 4), tflops=round(fl / ms / 1e9, 2)))
        tot_flops += fl; tot_ms += ms
        if r["cin"] % 8 == 0:
            mfma_flops += fl; mfma_ms += ms
        cur = out
    return layers, mfma_flops / mfma_ms / 1e9


def cpu_baseline(n_samples, kinds, budget_s=12.0):
    """The CPU oracle (a port of the reference path: serial per-candidate render + torch-CPU
    Cnn14) on a bounded sample: candidates are evaluated one at a time until ~budget_s of CPU
    work has been spent."""
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
    return {"value": round(n / dt, 4), "unit": "candidate-evals/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} candidates x 10 s stereo, same chain, serial oracle.evaluate "
            f"(C effects + torch-CPU Cnn14, {torch.get_num_threads()} threads of {os.cpu_count()} cpus), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pop-per-gpu", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # STITO_BENCH_BACKEND=gloo lets several ranks share one GPU (a functional check of the N > 1 path on a
    # 1-GPU box; RCCL refuses two ranks on one device).  The driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("STITO_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
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
    kinds = ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]
    plugins = E.make_plugins("bench5")
    D = sum(p["num_params"] for p in plugins.values())
    model = make_synthetic_param_model(seed=0, input_norm="minmax")
    x = synth_audio(1234, 2, n)[None]
    # target: the seed-4321 signal rendered through the same chain at w_target = default_rng(7)
    from st_ito.style_transfer import process_audio
    tgt = torch.from_numpy(process_audio(synth_audio(4321, 2, n).numpy(), np.random.default_rng(7).random(D), SR, plugins))[None]
    te = get_param_embeds(tgt, model, SR)
    ev = PopulationEvaluator(x, SR, plugins, model, te)
    P_total = args.pop_per_gpu * world
    es = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P_total, "seed": 42})

    def step():
        W = es.ask()
        lo, hi = shard_bounds(P_total, rank, world)
        loss, _, _ = ev.evaluate(W[lo:hi])
        f = gather_fitness(loss, P_total)
        es.tell(W, f.tolist())  # .tolist() = the device->host sync the optimiser needs anyway

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timing = rank == 0 and not args.no_roofline
    if timing:  # the library records HIP events around every MFMA conv launch of the timed steps
        _hip.check(_hip.lib().stito_conv_timing_enable(1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    conv_ms, conv_launches = ctypes.c_double(0.0), ctypes.c_int(0)
    if timing:
        _hip.check(_hip.lib().stito_conv_timing_enable(0))
        _hip.check(_hip.lib().stito_conv_timing_read(ctypes.byref(conv_ms), ctypes.byref(conv_launches)))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = {
        "metric": "candidate-evals/sec (pop x iters), 48 kHz 10 s stereo, 5-effect chain",
        "value": round(P_total * args.steps / dt, 3), "unit": "candidate-evals/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ES evaluate-population: pop={args.pop_per_gpu}/GPU ({P_total} total), 48 kHz stereo "
                   f"{args.seconds:g} s, chain EQ/comp/reverb/EQ/gain (D={D}), AFx-Rep Cnn14 (seeded random weights), "
                   "CMA-ES seed 42", "pop_per_gpu": args.pop_per_gpu, "n_samples": n, "chain": kinds,
                   "parallelism": f"population sharded over {world} GPU(s), fitness all-gather"},
    }
    if rank == 0:
        if not args.no_roofline:
            T = n // 1024 + 1
            # algorithmic conv FLOPs (direct-convolution count 2*9*cin*cout*H*W) of one step on this rank
            fl_step = sum(r["flops"] for r in conv_layer_table(T) if r["cin"] % 8 == 0) * 2 * args.pop_per_gpu
            n_l = max(conv_launches.value, 1)
            achieved = fl_step * args.steps / conv_ms.value / 1e9  # TFLOP/s over the timed region's conv launches
            streams_per_launch = min(2 * args.pop_per_gpu, model.max_streams_per_pass)
            layers, _ = conv_layer_times(model, streams_per_launch, T)
            out["roofline"] = {
                "bound": "mfma",
                "kernel": "k_conv_wino<*> / k_conv3x3<*>: the 11 f32-MFMA 3x3-conv launches of a trunk pass. achieved counts "
                          "direct-convolution FLOPs (2*9*cin*cout*H*W); the Winograd F(2x2,3x3) kernel issues 16/36 of those "
                          "MACs, so it can exceed 1.0 of the MFMA peak",
                "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                # the MFMA work actually issued (Winograd: 16 of the 36 counted MACs per 2x2 outputs) against the same peak
                "mfma_issued_frac": round(achieved * (16.0 / 36.0 if all(l["algo"] == "winograd" for l in layers if l["cin"] % 8 == 0) else 1.0)
                                          / MFMA_F32_PEAK_TFLOPS, 4),
                "traffic": pmc_traffic_per_launch(streams_per_launch), "traffic_unit": "HBM bytes per launch (PMC, "
                "profiles/round1_p_conv_pmc_traffic.json)",
                "flops_per_launch": fl_step * args.steps / n_l, "avg_launch_ms": round(conv_ms.value / n_l, 4),
                "launches_timed": conv_launches.value, "n_streams": streams_per_launch,
                # whole path (DSP + front end + trunk + host) against the same peak
                "end_to_end_frac": round(fl_step / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "layers": layers,
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, kinds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
