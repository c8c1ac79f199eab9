"""Inference-time optimisation (ES) of the reference, st_ito/style_transfer.py: `load_plugins`
(17-42), `process_audio` (45-115), `parameters_to_dict` (324-359), `savepop_to_disk` (362-396)
and `run_es` (399-692), with the evaluate-population step on the MI355X; `run_staged_es` is the fixed variant
of scripts/run_optim.py:39-234 on the same evaluate step.

Only the ES path is built; the baselines of the reference file (run_input, run_random,
run_rule_based, run_deepafx_st) are outside this build's scope.  `run_es_batch` is an extension
(BASELINE.json configs[2]): several (input, target) pairs optimised together, every iteration
evaluating all their populations in one GPU batch.
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import torch

from . import cmaes as cma
from . import engine


# ------- audio processing methods -------
def load_plugins(plugins: dict):
    """Instantiate every plugin of the dict and record its parameter layout (reference style_transfer.py:17-42).

    Fills plugin["instance"], plugin["parameter_names"] -- the (dead) "our_bypass" slot first, then the instance's
    parameters in their own order -- and plugin["num_params"]; returns (plugins, total number of slots, the initial raw
    values slot by slot), printing each parameter like the reference does."""
    init_params: List[float] = []
    for plugin_name, plugin in plugins.items():
        if "vst_filepath" in plugin:
            raise NotImplementedError("VST plugins (pedalboard.load_plugin) are not supported in this build")
        if "class_path" not in plugin:
            raise ValueError("Plugin must contain 'vst_filepath' or 'class_path'.")
        instance = plugin["class_path"]()
        raw = {name: prm.raw_value for name, prm in instance.parameters.items()}
        for name, value in raw.items():
            print(f"{plugin_name}: {name} = {value}")
        print()
        plugin["instance"] = instance
        plugin["parameter_names"] = ["our_bypass", *raw]
        plugin["num_params"] = 1 + len(raw)
        init_params += [0.0, *raw.values()]
    return plugins, len(init_params), init_params


def process_audio(x: np.ndarray, w: np.ndarray, sr: int, plugins: List[dict], normalize_stages: bool = False):
    """Process audio with plugins and provided parameters on [0, 1] (reference
    style_transfer.py:45-115).  x: (chs, num_samples) float32, w: (num_params,).  Rendered on the
    GPU by stito_render_population + stito_normalize_audio; normalize_stages (106-107) adds a joint
    peak normalisation after every plugin (STITO_FX_FLAG_NORMALIZE_AFTER)."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    if isinstance(w, torch.Tensor):
        w = w.detach().cpu().numpy()
    out = engine.process_audio_gpu(x, w, sr, plugins, normalize_stages=normalize_stages)
    # side effect of the reference's loop (71-87): the plugin instances end up holding the values they were rendered with (raw
    # slot for a free parameter, set_value of the fixed one) -- scripts/eval/eval_case_study.py:372-388 reads them back after a
    # dummy call.  The GPU render takes its values from w, so they are written here, on the host.
    _write_parameters(w, plugins)
    return out


def _write_parameters(w, plugins):
    """Walk w over the plugins' slots and leave every value in its Parameter; -> the values in the parameters' own units."""
    values = {}
    slot = iter(w)
    for plugin_name, plugin in plugins.items():
        mine = values.setdefault(plugin_name, {})
        fixed = plugin["fixed_parameters"]
        for name in plugin["parameter_names"]:
            raw = next(slot)  # every name consumes its slot, fixed or not
            if name == "our_bypass":
                mine[name] = raw
                continue
            prm = engine._instance_of(plugin).parameters[name]
            if name in fixed:
                prm.set_value(fixed[name])
            else:
                prm.raw_value = raw
            mine[name] = prm.get_value() if hasattr(prm, "get_value") else prm.raw_value
    return values


def parameters_to_dict(w: np.ndarray, plugins: List[dict]):
    """{plugin: {parameter: value in its own unit}} for a vector on [0, 1] (reference style_transfer.py:324-359).

    Like the reference this WRITES the values into the plugin instances on the way (raw value for a free slot,
    `set_value` of the fixed value for a fixed one); "our_bypass" is reported as the raw slot."""
    return _write_parameters(w, plugins)


def savepop_to_disk(iteration, fvals, output_embeds, output_audios, run_dir: str, sample_rate: int, first: int = 0):
    """reference style_transfer.py:362-396: one wav per candidate, sorted by fitness.

    `fvals` is the full population's fitness; `output_audios` holds candidates [first, first + len)
    of it -- under torch.distributed every rank passes its own shard and writes only its own
    candidates' files, named by their rank in the global ordering, so the directory ends up with the
    same files a single process writes (no audio is communicated).

    STATED DEVIATION, one switch away from parity: the reference zips (fvals, output_audios, output_embeds) and run_es hands
    it the embedding DICT, whose iteration yields its keys -- so the reference writes only as many files as the dict has
    entries (two for AFx-Rep: candidates 0 and 1 of the population, ranked among themselves).  The default here writes the
    whole population, which is what the flag promises; STITO_SAVEPOP_REFERENCE=1 reproduces the reference's files exactly
    (tests/test_host_logic.py against a fixture the reference's own function produced)."""
    from .audio_io import save_wav

    pop_dir = os.path.join(run_dir, f"pop_{iteration}")
    os.makedirs(pop_dir, exist_ok=True)
    members = range(len(fvals))
    if os.environ.get("STITO_SAVEPOP_REFERENCE") == "1":
        members = range(min(len(fvals), len(output_embeds)))   # zip() stops at the shortest: the dict's keys
    order = sorted(members, key=lambda i: fvals[i])
    for idx, i in enumerate(order):
        if not first <= i < first + len(output_audios):
            continue
        audio = output_audios[i - first]
        audio = audio / torch.max(torch.abs(audio)).clamp(min=1e-8)
        save_wav(os.path.join(pop_dir, f"output_audio_pop_{idx}_fval_{fvals[i]:0.4e}.wav"), audio.cpu(), sample_rate)


# ----------- Evolutionary Strategies ------------
def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def shard_bounds(popsize: int, rank: int, world: int):
    """Candidates [lo, hi) of the ask() batch owned by `rank` (contiguous, near-equal shards)."""
    base, rem = divmod(popsize, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_fitness(local: torch.Tensor, popsize: int) -> torch.Tensor:
    """All-gather the per-rank fitness shards into candidate order (RCCL when the tensors are on
    the GPU, gloo on CPU).  Shards may differ by one element, so pad to the largest."""
    dist, rank, world = _dist_info()
    # a one-rank group skips the collective unless STITO_FORCE_COLLECTIVE=1 (tests/test_gpu_es.py pushes a step through RCCL's
    # communicator setup and all_gather_into_tensor on the one GPU a test box has)
    if world == 1 and not (dist is not None and os.environ.get("STITO_FORCE_COLLECTIVE") == "1"):
        return local
    per = (popsize + world - 1) // world
    buf = torch.full((per,), float("inf"), dtype=local.dtype, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty((world, per), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), buf)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(popsize, r, world)
        parts.append(out[r, : hi - lo])
    return torch.cat(parts)


def sharded_evaluate(W, eval_local, while_waiting=None):
    """Evaluate a population across the ranks of the default process group.

    `eval_local(W_shard) -> (loss tensor (n_local,), embeds, audios)` is called with this rank's
    contiguous shard of W; the per-rank fitness vectors are all-gathered into candidate order, so
    every rank returns the same full fitness list (and its own shard's embeds/audios).  `while_waiting()` runs on the host
    after the shard's GPU work has been queued and before the fitness download waits for it."""
    _, rank, world = _dist_info()
    P = len(W)
    lo, hi = shard_bounds(P, rank, world)
    loss, embeds, audios = eval_local(W[lo:hi])
    if while_waiting is not None:
        while_waiting()
    return gather_fitness(loss, P).tolist(), embeds, audios


def run_es(
    input_audio: torch.Tensor,
    target_audio: torch.Tensor,
    sample_rate: int,
    plugins: List[dict],
    model: torch.nn.Module,
    embed_func: callable,
    content_model: torch.nn.Module = None,
    content_embed_func: callable = None,
    max_iters: int = 100,
    w0: torch.Tensor = None,
    find_w0: bool = True,
    sigma0: float = 0.1,
    distance: str = "cosine",
    random_crop: bool = False,
    popsize: int = 32,
    parallel: bool = False,
    dropout: float = 0.0,
    savepop: bool = False,
    run_dir: str = ".",
    seed: int = None,
    early_stop: bool = True,
    *args,
    **kwargs,
):
    """Run CMA-ES optimization to find the best parameters (reference style_transfer.py:399-692).

    Same arguments and result dict as the reference.  `parallel` selects the reference's pool branch (499-502), whose only
    observable difference is the length policy: the candidates are rendered from the input as it is (no zero padding to
    262144 samples, no random crop); the whole population is rendered on the GPU at once either way.  Extensions: `seed` (CMA-ES + find_w0 RNG; the
    reference is unseeded) and `early_stop` (False disables the break of lines 655-670 for fixed
    work benchmarks).  Under torch.distributed each rank evaluates a contiguous shard of the
    population and the fitness scalars are all-gathered; the CMA-ES state is replicated."""
    if distance != "cosine":
        raise ValueError(f"Unknown distance: {distance}")
    if content_model is not None and content_embed_func is None:
        raise ValueError("content_model needs a content_embed_func")
    total_num_params = sum([plugin["num_params"] for plugin in plugins.values()])
    bs, chs, seq_len = input_audio.shape
    dist, rank, world = _dist_info()
    if world > 1 and seed is None:
        # every rank steps a replica of the CMA-ES state and slices the same ask() batch: the replicas (and the
        # find_w0 / random-crop draws) must be identical, so an unseeded run agrees on rank 0's draw of a seed
        box = [int(np.random.SeedSequence().generate_state(1)[0] & 0x7FFFFFFF) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    rng = np.random.RandomState(seed) if seed is not None else np.random

    # peak normalize (in place like the reference, 452-453)
    input_audio /= torch.max(torch.abs(input_audio)).clamp(min=1e-8)
    target_audio /= torch.max(torch.abs(target_audio)).clamp(min=1e-8)

    # compute target embedding (only once)
    target_embed = embed_func(target_audio, model, sample_rate)

    # content branch (468-473, 537-542, 560-568): a second embedding of the same rendered audio, scored against the TARGET's
    # content embedding with twice the weight.  It rides on the generic-metric path: one embedding dict with the content
    # entries under a prefix, the mean over all entries with weight 2 on those.
    eval_embed_func, eval_targets, entry_weights = embed_func, target_embed, None
    _CP = "__content__:"
    if content_model is not None:
        target_content_embeds = content_embed_func(target_audio, content_model, sample_rate)
        eval_targets = dict(target_embed)
        eval_targets.update({_CP + k: v for k, v in target_content_embeds.items()})
        entry_weights = {_CP + k: 2.0 for k in target_content_embeds}

        def eval_embed_func(x, m, sr):
            e = dict(embed_func(x, m, sr))
            e.update({_CP + k: v for k, v in content_embed_func(x, content_model, sr).items()})
            return e

    # (run_optim.py:608 passes normalize_stages=...; the reference's run_es swallows it in **kwargs and its evaluate
    # never forwards it to process_audio, so the population is rendered without per-stage normalisation here too)
    evaluator = engine.PopulationEvaluator(input_audio, sample_rate, plugins, model, eval_targets, embed_func=eval_embed_func,
                                           entry_weights=entry_weights)
    if evaluator.ndims != total_num_params:
        raise ValueError(f"plugins declare {total_num_params} params, chain consumes {evaluator.ndims}")

    def evaluate(W, dropout: float = 0.0, want_audio: bool = False, while_waiting=None):
        """GPU replacement of the reference's evaluate closure (474-573)."""
        out = sharded_evaluate(W, lambda Ws: evaluator.evaluate(Ws, random_crop=random_crop, rng=rng,
                                                                want_audio=want_audio, dropout=dropout, parallel=parallel),
                               while_waiting)
        warn = evaluator.nan_warning()  # after the fitness download: no extra synchronisation
        if warn:
            print(warn)
        if content_model is not None:  # the reference hands back the style embeddings only (573)
            out = (out[0], {k: v for k, v in out[1].items() if not k.startswith(_CP)}, out[2])
        return out

    # setup CMA-ES
    if find_w0:
        print("Finding the best w0...")
        tmp_w0s = [rng.rand(total_num_params) for _ in range(popsize)]
        fvals, output_embeds, output_audios = evaluate(tmp_w0s, dropout=dropout, want_audio=savepop)
        print(fvals)
        w0 = tmp_w0s[int(np.argmin(fvals))]
        if savepop:
            savepop_to_disk(-1, fvals, output_embeds, output_audios, run_dir, sample_rate,
                            first=shard_bounds(len(fvals), rank, world)[0])
    else:
        if w0 is None:
            w0 = np.ones(total_num_params) * 0.5
        else:
            w0 = w0.numpy() if isinstance(w0, torch.Tensor) else np.asarray(w0)

    init_param_dict = parameters_to_dict(w0, plugins)
    print(init_param_dict)

    opts = {"bounds": [0, 1], "popsize": popsize}
    if seed is not None:
        opts["seed"] = seed
    es = cma.CMAEvolutionStrategy(w0, sigma0, opts)

    fval_history = []
    wopt_history = []
    iters_without_improvement = 0
    n_evals = popsize if find_w0 else 0

    for iteration in range(max_iters):
        W = es.ask()
        # (the next generation's normal deviates are drawn on the host while the GPU evaluates this one)
        fvals, output_embeds, output_audios = evaluate(
            W, dropout=(dropout if (iteration + 1) < max_iters else 0.0), want_audio=savepop,
            while_waiting=getattr(es, "prefetch", None))
        n_evals += len(W)

        # save best (pre-tell result, like the reference: index 0 is (None, inf))
        wopt_history.append(es.result[0])
        fval_history.append(es.result[1])

        if savepop:
            savepop_to_disk(iteration, fvals, output_embeds, output_audios, run_dir, sample_rate,
                            first=shard_bounds(len(fvals), rank, world)[0])
        es.tell(W, fvals)
        if rank == 0:
            es.disp()

        # early stop (reference 655-670): an iteration counts as stale unless its best candidate beats the best value on
        # record (the pre-tell history) by more than 0.01; the first iteration never counts
        stale = iteration > 0 and min(fvals) - min(fval_history) > -0.01
        iters_without_improvement = iters_without_improvement + 1 if stale else 0
        if stale and rank == 0:
            print(f"Solution has not improved for {iters_without_improvement} iterations.")
        if early_stop and iters_without_improvement > 10:
            print("Stopping early due to no improvement.")
            break

    wopt = es.result[0]
    fopt = es.result[1]

    # render the current solution on the full (un-padded) input, like the reference (676-678)
    output_audio = torch.from_numpy(process_audio(input_audio.squeeze(0).cpu().numpy(), wopt, sample_rate, plugins))
    param_dict = parameters_to_dict(wopt, plugins)
    return {
        "output_audio": output_audio,
        "params": param_dict,
        "fopt": fopt,
        "wopt": wopt,
        "fval_history": fval_history,
        "wopt_history": wopt_history,
        "num_evals": n_evals,
    }


def run_staged_es(
    input_audio: torch.Tensor,
    target_audio: torch.Tensor,
    sample_rate: int,
    plugins: List[dict],
    model: torch.nn.Module,
    embed_func: callable,
    normalization: str = "peak",
    max_iters: int = 100,
    w0: torch.Tensor = None,
    popsize: int = 10,
    sigma0: float = 0.1,
    distance: str = "cosine",
    parallel: bool = False,
    save_pop: bool = False,
    savepop: bool = False,
    run_dir: str = ".",
    seed: int = None,
    *args,
    **kwargs,
):
    """Stage-wise CMA-ES (reference scripts/run_optim.py:39-234, `--staged`): stage k optimises ONLY the
    parameters of plugin k on the sub-chain plugins[0..k], with the earlier plugins held at their stage optima;
    every stage starts from 0.5 in its own dimensions and gets max_iters // len(plugins) iterations (147-188).

    The reference variant cannot run as written (it calls an undefined `parameters_to_dict`, writes to a global
    `run_dir`, takes the cosine of the embedding *dict*, adds a fourth dimension to the target, and returns a
    tuple where its caller reads a dict).  This is the fixed variant on the GPU evaluate step: the stage's
    candidates are the reference's composed vectors `[wopt_overall, w]` (161-166), rendered and scored by
    PopulationEvaluator like run_es does (same length policy, same loss: mean over the embed_func dict of
    -cosine), input and target peak-normalised in place like run_es (452-453), population sharded over the
    ranks like run_es.  Returns run_es's dict; fval_history / wopt_history hold the stage-local best after
    every tell (181-185), `stage_wopts` the per-stage optima.  Stage k's CMA-ES is seeded with seed + k."""
    if distance != "cosine":
        raise ValueError(f"Unknown distance: {distance}")
    savepop = bool(savepop or save_pop)
    dist, rank, world = _dist_info()
    if world > 1 and seed is None:
        box = [int(np.random.SeedSequence().generate_state(1)[0] & 0x7FFFFFFF) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    input_audio /= torch.max(torch.abs(input_audio)).clamp(min=1e-8)
    target_audio /= torch.max(torch.abs(target_audio)).clamp(min=1e-8)
    target_embed = embed_func(target_audio, model, sample_rate)

    names = list(plugins.keys())
    iters_per_stage = max_iters // len(plugins)
    wopt_overall, fopt = None, float("inf")
    fval_history, wopt_history, stage_wopts = [], [], []
    n_evals = 0
    output_audio = None
    for stage_idx in range(len(plugins)):
        stage_plugins = {k: plugins[k] for k in names[: stage_idx + 1]}
        print(f"Optimizing stage {stage_idx} ({list(stage_plugins.keys())})")
        total_num_params = sum(p["num_params"] for p in stage_plugins.values())
        n_stage = plugins[names[stage_idx]]["num_params"]
        evaluator = engine.PopulationEvaluator(input_audio, sample_rate, stage_plugins, model, target_embed, embed_func=embed_func)
        if evaluator.ndims != total_num_params:
            raise ValueError(f"plugins declare {total_num_params} params, chain consumes {evaluator.ndims}")
        opts = {"bounds": [0, 1], "popsize": popsize}
        if seed is not None:
            opts["seed"] = seed + stage_idx
        es = cma.CMAEvolutionStrategy(np.ones(n_stage) * 0.5, sigma0, opts)
        for iteration in range(iters_per_stage):
            W = es.ask()
            if stage_idx > 0:
                W_stage = [np.concatenate([wopt_overall, w]) for w in W]
            else:
                W_stage = W
            fvals, _, output_audios = sharded_evaluate(W_stage, lambda Ws: evaluator.evaluate(Ws, want_audio=savepop))
            n_evals += len(W)
            es.tell(W, fvals)
            if rank == 0:
                es.disp()
            if savepop:  # 171-179: one file per candidate, overwritten every iteration of the stage
                from .audio_io import save_wav

                lo = shard_bounds(len(fvals), rank, world)[0]
                for j, audio in enumerate(output_audios):
                    audio = audio / torch.max(torch.abs(audio)).clamp(min=1e-8)
                    save_wav(os.path.join(run_dir, f"output_audio_stage_{stage_idx}_pop_{lo + j}_fval_{fvals[lo + j]:0.3f}.wav"),
                             audio.cpu(), sample_rate)
            fval_history.append(es.result[1])
            wopt_history.append(es.result[0])
        wopt, fopt = es.result[0], es.result[1]
        if wopt is None:  # max_iters < len(plugins): no iteration ran
            wopt = np.ones(n_stage) * 0.5
        stage_wopts.append(wopt)
        wopt_overall = wopt if wopt_overall is None else np.concatenate([wopt_overall, wopt])
        output_audio = torch.from_numpy(process_audio(input_audio.squeeze(0).cpu().numpy(), wopt_overall, sample_rate, stage_plugins))
        if rank == 0 and run_dir is not None and os.path.isdir(run_dir):
            from .audio_io import save_wav

            save_wav(os.path.join(run_dir, f"output_audio_stage_{stage_idx}.wav"), output_audio, sample_rate)
    param_dict = parameters_to_dict(wopt_overall, plugins)
    return {
        "output_audio": output_audio,
        "params": param_dict,
        "fopt": fopt,
        "wopt": wopt_overall,
        "fval_history": fval_history,
        "wopt_history": wopt_history,
        "stage_wopts": stage_wopts,
        "num_evals": n_evals,
    }


def run_es_batch(
    input_audios: torch.Tensor,
    target_audios: torch.Tensor,
    sample_rate: int,
    plugins: List[dict],
    model: torch.nn.Module,
    embed_func: callable,
    max_iters: int = 100,
    sigma0: float = 0.1,
    popsize: int = 32,
    random_crop: bool = False,
    seed: int = None,
    early_stop: bool = True,
):
    """ES over B independent (input, target) pairs at once -- BASELINE.json configs[2].

    The reference optimises its examples one after the other (scripts/eval/eval_pst.py:691-765)
    and its evaluate closure assumes one input (style_transfer.py:520).  Here every pair keeps its
    own CMA-ES state (seed + pair index) and each iteration evaluates the B populations, stacked
    pair-major as (B * popsize, D), in one pass over the GPU; pair b's candidates read input b and
    are scored against target b.  A pair's trajectory is bitwise the one `run_es(find_w0=False,
    seed=seed + b)` produces for it alone.  Pairs that stop early (same rule as run_es, lines
    655-670) keep being evaluated with their last population but are no longer told.

    input_audios / target_audios: (B, chs, seq_len), all pairs the same length and channel count.
    Under torch.distributed the PAIRS are sharded over the ranks (no collective until
    the final gather); every rank returns the full list of B result dicts."""
    if input_audios.dim() != 3 or target_audios.dim() != 3 or input_audios.shape[0] != target_audios.shape[0]:
        raise ValueError("input_audios and target_audios must be (B, chs, seq_len) with the same B")
    dist, rank, world = _dist_info()
    B_all = input_audios.shape[0]
    lo, hi = shard_bounds(B_all, rank, world)
    results = [None] * B_all
    if hi > lo:
        xs = input_audios[lo:hi].clone()
        ts = target_audios[lo:hi].clone()
        B = hi - lo
        total_num_params = sum([plugin["num_params"] for plugin in plugins.values()])
        # peak normalise each pair on its own (run_es 452-453)
        xs /= xs.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-8)
        ts /= ts.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-8)
        target_embed = embed_func(ts, model, sample_rate)
        evaluator = engine.PopulationEvaluator(xs, sample_rate, plugins, model, target_embed, embed_func=embed_func)
        if evaluator.ndims != total_num_params:
            raise ValueError(f"plugins declare {total_num_params} params, chain consumes {evaluator.ndims}")
        rng = np.random.RandomState(seed) if seed is not None else np.random
        states = []
        for b in range(B):
            opts = {"bounds": [0, 1], "popsize": popsize}
            if seed is not None:
                opts["seed"] = seed + lo + b
            states.append(dict(es=cma.CMAEvolutionStrategy(np.ones(total_num_params) * 0.5, sigma0, opts), fval_history=[],
                               wopt_history=[], stale=0, active=True, last_W=None, n_evals=0))
        for iteration in range(max_iters):
            if not any(st["active"] for st in states):
                break
            Ws = []
            for st in states:
                if st["active"]:
                    st["last_W"] = st["es"].ask()
                Ws.append(st["last_W"])
            loss, _, _ = evaluator.evaluate(np.concatenate([np.asarray(W) for W in Ws], 0), random_crop=random_crop, rng=rng)
            fv = loss.tolist()
            for b, st in enumerate(states):
                if not st["active"]:
                    continue
                fvals = fv[b * popsize:(b + 1) * popsize]
                st["n_evals"] += popsize
                st["wopt_history"].append(st["es"].result[0])
                st["fval_history"].append(st["es"].result[1])
                st["es"].tell(st["last_W"], fvals)
                fval_delta = (min(fvals) - min(st["fval_history"])) if iteration > 0 else -0.02
                st["stale"] = st["stale"] + 1 if fval_delta > -0.01 else 0
                if early_stop and st["stale"] > 10:
                    st["active"] = False
        for b, st in enumerate(states):
            wopt, fopt = st["es"].result[0], st["es"].result[1]
            out = torch.from_numpy(process_audio(xs[b].cpu().numpy(), wopt, sample_rate, plugins))
            results[lo + b] = {"output_audio": out, "params": parameters_to_dict(wopt, plugins), "fopt": fopt, "wopt": wopt,
                               "fval_history": st["fval_history"], "wopt_history": st["wopt_history"],
                               "num_evals": st["n_evals"]}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, [(i, r) for i, r in enumerate(results) if r is not None])
        for part in gathered:
            for i, r in part:
                results[i] = r
    return results
