#!/usr/bin/env python
"""scripts/run_optim.py of the reference (lines 300-645) for the ES path on MI355X.

    python scripts/run_optim.py input.wav target.wav --algorithm es --effect-type basic --metric param

Same flags as the reference; the ones whose code paths are outside this build raise a clear
error (--effect-type vst, --algorithm autodiff, --metric clap); --staged runs the fixed run_staged_es
(st_ito.style_transfer).  Extensions:
--target (README.md:18 spelling), --seed, --chain, --ckpt, --synthetic, --no-early-stop,
--no-find-w0.  Multi-GPU: launch with torch.distributed.run, one rank per GPU; the population
is sharded and fitness all-gathered (RCCL).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from st_ito import effects  # noqa: E402
from st_ito.audio_io import load_wav, resample, save_wav  # noqa: E402
from st_ito.style_transfer import process_audio, run_es, run_staged_es  # noqa: E402
from st_ito.utils import get_param_embeds, load_param_model, make_synthetic_param_model  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("input", type=str)
    parser.add_argument("target_pos", type=str, nargs="?", default=None, metavar="target")
    parser.add_argument("--target", type=str, default=None)
    parser.add_argument("--max-iters", type=int, default=300)
    parser.add_argument("--popsize", type=int, default=32)
    parser.add_argument("--max-length", type=int, default=262144)
    parser.add_argument("--staged", action="store_true")
    parser.add_argument("--savepop", action="store_true")
    parser.add_argument("--normalize-stages", action="store_true")
    parser.add_argument("--use-gpu", action="store_true")
    parser.add_argument("--parallel", action="store_true")
    parser.add_argument("--effect-type", type=str, default="vst", choices=["vst", "basic"])
    parser.add_argument("--algorithm", type=str, default="es", choices=["es", "autodiff"])
    parser.add_argument("--dropout", type=float, default=0.0)
    parser.add_argument("--metric", type=str, default="param", choices=["param", "clap"])
    # extensions
    parser.add_argument("--seed", type=int, default=None)
    parser.add_argument("--chain", type=str, default=None, choices=sorted(effects.BASIC_CHAINS))
    parser.add_argument("--ckpt", type=str, default=None)
    parser.add_argument("--synthetic", action="store_true", help="seeded random AFx-Rep weights (no checkpoint)")
    parser.add_argument("--no-early-stop", action="store_true")
    parser.add_argument("--no-find-w0", action="store_true")
    parser.add_argument("--output-dir", type=str, default="output/optim")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    target = args.target or args.target_pos
    sample_rate = 48000
    if args.algorithm != "es":
        raise NotImplementedError("--algorithm autodiff (run_optim.py:237-297) is outside this build: ES path only")
    if args.metric != "param":
        raise NotImplementedError("--metric clap needs laion_clap + downloaded weights; only the AFx-Rep metric is built")
    if args.effect_type == "vst" and args.chain is None:
        raise NotImplementedError("--effect-type vst needs pedalboard VST3 hosting; use --effect-type basic")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    # plugins exactly as run_optim.py:375-437 builds them (no bypass slot)
    plugins = effects.make_plugins(args.chain or "basic", with_bypass=False)
    total_num_params = 0
    init_params = []
    for plugin_name, plugin in plugins.items():
        for name, parameter in plugin["instance"].parameters.items():
            if rank == 0:
                print(f"{plugin_name}: {name} = {parameter.raw_value}")
            init_params.append(parameter.raw_value)
        total_num_params += plugin["num_params"]
    w0 = torch.tensor(init_params, dtype=torch.float32)

    input_audio, input_sr = load_wav(args.input)
    input_name = os.path.basename(args.input).replace(".wav", "")
    input_audio = resample(input_audio, input_sr, sample_rate)
    if target is None:
        # the reference's unreachable synthetic-target branch (run_optim.py:452-521): render the
        # input through the chain at fixed parameters
        w_target = np.random.RandomState(0 if args.seed is None else args.seed).rand(total_num_params)
        target_audio = torch.from_numpy(process_audio(input_audio.numpy(), w_target, sample_rate, plugins))
        target_name = "synthetic_target"
    else:
        target_audio, target_sr = load_wav(target)
        target_name = os.path.basename(target).replace(".wav", "")
        target_audio = resample(target_audio, target_sr, sample_rate)

    input_audio = input_audio[:, : args.max_length].contiguous()
    target_audio = target_audio[:, : args.max_length].contiguous()

    run_name = f"{input_name}_to_{target_name}_{args.algorithm}"
    run_dir = os.path.join(args.output_dir, run_name)
    os.makedirs(run_dir, exist_ok=True)

    if args.synthetic:
        model = make_synthetic_param_model(seed=0)
    else:
        model = load_param_model(ckpt_path=args.ckpt, use_gpu=args.use_gpu)
    embed_func = get_param_embeds

    if rank == 0:
        save_wav(os.path.join(run_dir, "input_audio.wav"), input_audio, sample_rate)
    target_audio /= torch.max(torch.abs(target_audio)).clamp(min=1e-8)
    if rank == 0:
        save_wav(os.path.join(run_dir, "target_audio.wav"), target_audio, sample_rate)

    sigma0 = 0.33
    print(f"Running ES with sigma0 = {sigma0}")
    es_func = run_staged_es if args.staged else run_es  # run_optim.py:582
    result = es_func(
        input_audio.unsqueeze(0), target_audio.unsqueeze(0), sample_rate, plugins, model, embed_func,
        max_iters=args.max_iters, popsize=args.popsize, w0=w0, find_w0=not args.no_find_w0, sigma0=sigma0,
        distance="cosine", parallel=args.parallel, dropout=args.dropout, savepop=args.savepop,
        normalize_stages=args.normalize_stages, run_dir=run_dir, seed=args.seed,
        early_stop=not args.no_early_stop,
    )
    output_audio = result["output_audio"]
    if rank == 0:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, axs = plt.subplots(1, 2, figsize=(10, 5))
            axs[0].plot(result["fval_history"], label=f"sigma0={sigma0:0.2f}")
            axs[0].set_xlabel("Iteration"); axs[0].set_ylabel("Distance"); axs[0].legend()
            plt.savefig(os.path.join(run_dir, "plot.png"), dpi=150)
        except Exception as e:  # plotting is optional
            print(f"(plot skipped: {e})")
        output_audio /= torch.max(torch.abs(output_audio)).clamp(min=1e-8)
        save_wav(os.path.join(run_dir, f"output_audio_sigma={sigma0:0.2f}.wav"), output_audio.squeeze(0), sample_rate)
        with open(os.path.join(run_dir, f"parameters_sigma={sigma0:0.2f}.json"), "w") as f:
            json.dump(result["params"], f, indent=4)
        print(f"fopt = {result['fopt']:.6f} after {result['num_evals']} candidate evaluations -> {run_dir}")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
