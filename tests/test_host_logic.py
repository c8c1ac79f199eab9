"""CPU tests of the host side: C-ABI surface, chain compilation, CMA-ES, population sharding
(world_size-2 gloo), and the no-fallback rule."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from st_ito import _hip
    hdr = open(os.path.join(ROOT, "include", "stito_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(stito_\w+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = _hip.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/stito_hip.h but not exported"
    assert declared == set(_hip.SIGNATURES), declared ^ set(_hip.SIGNATURES)
    assert lib.stito_version() == 10
    for retired in (6, 7):   # the split-precision experiments of rounds 3 - 4: numbers reserved, nothing behind them
        assert lib.stito_conv3x3_supported(2, 32, 32, 64, 64, 0, retired) == 0
        assert lib.stito_cnn14_packed_conv_floats(64, 64, retired) == 0
    for kind, n in enumerate([18, 4, 2, 3, 4, 1, 25, 5]):
        assert lib.stito_fx_num_params(kind) == n
    assert lib.stito_fx_num_params(99) < 0


def test_compile_chain_offsets_bypass_fixed():
    from st_ito import effects as E, _hip
    from st_ito.engine import compile_chain
    lib = _hip.lib()
    pl = E.make_plugins("bench5")
    d, n = compile_chain(pl)
    assert n == 45 == lib.stito_chain_num_dims(d, 5)
    assert [x.w_offset for x in d][:5] == [0, 18, 22, 26, 44]
    assert lib.stito_chain_out_channels(d, 5, 1) == 2 and lib.stito_chain_out_channels(d, 5, 2) == 2
    pl = E.make_plugins("basic", with_bypass=True)  # load_plugins layout: +1 dead slot per plugin
    d, n = compile_chain(pl)
    assert n == 31 + 5 and all(x.has_bypass == 1 for x in list(d)[:5])
    pl = E.make_plugins("eq-comp")
    pl["Compressor"]["fixed_parameters"] = {"ratio": 4.0}
    d, n = compile_chain(pl)
    assert n == 22 and d[1].fixed_mask == 0b10
    assert abs(d[1].fixed_raw[1] - (4.0 - 1.0) / 19.0) < 1e-15
    assert lib.stito_chain_out_channels(d, 2, 1) == 1  # mono stays mono (BASELINE config 0)
    with pytest.raises(AssertionError):  # Parameter.set_value range assert (effects.py:792)
        pl["Compressor"]["fixed_parameters"] = {"ratio": 40.0}
        compile_chain(pl)
    with pytest.raises(ValueError):
        compile_chain({"x": {"num_channels": 1, "fixed_parameters": {}}})
    with pytest.raises(NotImplementedError):
        compile_chain({"x": {"vst_filepath": "a.vst3", "num_channels": 1, "fixed_parameters": {}}})
    ws = lib.stito_render_workspace_bytes(compile_chain(E.make_plugins("bench5"))[0], 5, 2, 480000, 256)
    assert ws > 256 * 2 * 480000 * 4  # envelope buffer for the compressor


def test_parameter_protocol_and_load_plugins(capsys):
    from st_ito import effects as E
    from st_ito.style_transfer import load_plugins, parameters_to_dict
    p = E.Parameter(80.0, 20.0, 4000.0)
    assert p.raw_value == (80.0 - 20.0) / 3980.0 and p.get_value() == p.raw_value * 3980.0 + 20.0
    with pytest.raises(AssertionError):
        p.set_value(1.0)
    plugins = {"ParametricEQ": {"class_path": E.BasicParametricEQ, "num_params": None, "num_channels": 1,
                                "fixed_parameters": {}}}
    plugins, total, init = load_plugins(plugins)
    assert total == 19 and init[0] == 0.0 and plugins["ParametricEQ"]["parameter_names"][0] == "our_bypass"
    w = np.linspace(0.1, 0.9, 19)
    d = parameters_to_dict(w, plugins)
    assert d["ParametricEQ"]["our_bypass"] == w[0]
    assert d["ParametricEQ"]["low_shelf_gain_db"] == w[1] * 48.0 - 24.0
    ch = E.BasicChorus()   # reference effects.py:962-985: five declared parameters (rate_hz among them)
    assert list(ch.parameters) == ["rate_hz", "centre_delay_ms", "depth", "feedback", "mix"]
    assert ch.parameters["centre_delay_ms"].get_value() == pytest.approx(7.0) and ch.KIND == 7


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never route through the oracle."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from st_ito import effects as E, _hip
    from st_ito.style_transfer import process_audio
    from st_ito.utils import make_synthetic_param_model, get_param_embeds
    with pytest.raises(_hip.StitoError):
        process_audio(np.zeros((1, 4096), np.float32), np.full(18, 0.5), 48000, E.make_plugins("eq"))
    with pytest.raises(_hip.StitoError):
        E.BasicCompressor().process(np.zeros((1, 4096), np.float32), 48000)
    for mod in ("engine", "utils", "style_transfer", "effects", "cmaes", "models/panns", "_hip"):
        src = open(os.path.join(ROOT, "st-ito_amd", "st_ito", mod + ".py")).read()
        assert "st_ito_oracle" not in src and "oracle" not in src.replace("oracle's deterministic fill", ""), mod


def test_bound_transform():
    from st_ito.cmaes import BoundTransform
    bt = BoundTransform(0.0, 1.0)
    assert bt.al == 0.05 and bt.au == 0.1
    y = np.linspace(-3, 4, 2001)
    x = bt(y)
    assert x.min() >= 0.0 and x.max() <= 1.0
    inner = (y >= 0.05) & (y <= 0.9)
    np.testing.assert_allclose(x[inner], y[inner], atol=1e-15)  # identity in the interior
    assert abs(bt(np.array([-0.05]))[0]) < 1e-15 and abs(bt(np.array([1.1]))[0] - 1.0) < 1e-15
    assert np.abs(np.diff(x)).max() < 2 * (y[1] - y[0])  # continuous, slope <= 1
    # inverse (pycma maps x0 through it: the first distribution is centred on w0 itself)
    xs = np.concatenate([np.linspace(0, 1, 1001), [0.0, 0.0125, 0.05, 0.9, 0.999, 1.0]])
    np.testing.assert_allclose(bt(bt.inverse(xs)), xs, atol=1e-15)
    assert bt.inverse(np.array([0.0]))[0] == -0.05 and bt.inverse(np.array([1.0]))[0] == 1.1
    from st_ito.cmaes import CMAEvolutionStrategy
    w0 = np.array([0.0, 0.01, 0.5, 0.97, 1.0])
    es = CMAEvolutionStrategy(w0, 0.2, {"bounds": [0, 1], "popsize": 8, "seed": 0})
    np.testing.assert_allclose(es.result[5], w0, atol=1e-15)   # xmean (phenotype) == w0, also next to the bounds


def test_cmaes_deterministic_and_converges():
    from st_ito.cmaes import CMAEvolutionStrategy
    f = lambda x: float(np.sum((x - np.linspace(0.1, 0.9, 12)) ** 2))

    def run(seed):
        es = CMAEvolutionStrategy(np.full(12, 0.5), 0.33, {"bounds": [0, 1], "popsize": 24, "seed": seed})
        assert es.result[0] is None and es.result[1] == float("inf")  # pre-tell sentinel, like pycma
        for _ in range(120):
            X = es.ask()
            assert all((x >= 0).all() and (x <= 1).all() for x in X) and len(X) == 24
            es.tell(X, [f(x) for x in X])
        return es.result
    es = CMAEvolutionStrategy(np.full(12, 0.5), 0.33, {"bounds": [0, 1], "popsize": 24, "seed": 0})
    # pycma defaults: active CMA (negative weights for the worse half), c_sigma = (mueff + 2) / (N + mueff + 3)
    assert abs(es.weights.sum() - 1.0) < 1e-12 and (es.weights_all[12:] < 0).all()
    neg_sum = -es.weights_all[12:].sum()
    assert neg_sum <= 1 + es.c1 / es.cmu + 1e-12 and neg_sum <= (1 - es.c1 - es.cmu) / (12 * es.cmu) + 1e-12
    assert es.cs == (es.mueff + 2) / (12 + es.mueff + 3)
    a, b, c = run(3), run(3), run(4)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1] and not np.array_equal(a[0], c[0])
    assert a[1] < 1e-7
    # rank-based: any strictly monotone transform of the fitness gives the same trajectory
    es1 = CMAEvolutionStrategy(np.full(5, 0.5), 0.3, {"bounds": [0, 1], "popsize": 10, "seed": 1})
    es2 = CMAEvolutionStrategy(np.full(5, 0.5), 0.3, {"bounds": [0, 1], "popsize": 10, "seed": 1})
    for _ in range(20):
        X1, X2 = es1.ask(), es2.ask()
        es1.tell(X1, [np.sum(x ** 2) for x in X1])
        es2.tell(X2, [np.exp(np.sum(x ** 2)) for x in X2])
    np.testing.assert_array_equal(es1.result[0], es2.result[0])


def test_cmaes_active_matches_published_evaluation_counts():
    """Known answer for the optimiser itself: on the 10-D ellipsoid (condition 1e6, x0 = 3, sigma0 = 1, default
    popsize 10) CMA-ES needs about 5 700 evaluations to reach 1e-9 and active CMA about 4 200 (Hansen's tutorial /
    pycma's own figures); the update rule here must land in those ranges, active ahead of plain."""
    from st_ito.cmaes import CMAEvolutionStrategy
    N = 10
    f = lambda x: float(np.sum((10 ** (6 * np.arange(N) / (N - 1))) * x ** 2))
    evals = {}
    for active in (False, True):
        n = []
        for seed in range(3):
            es = CMAEvolutionStrategy(np.full(N, 3.0), 1.0, {"seed": seed, "CMA_active": active})
            assert es.popsize == 10 and es.boundary is None
            while es.result[1] > 1e-9 and es.countiter < 2000:
                X = es.ask()
                es.tell(X, [f(x) for x in X])
            n.append(es.counteval)
        evals[active] = np.mean(n)
    assert 4800 < evals[False] < 7000 and 3400 < evals[True] < 5200 and evals[True] < 0.85 * evals[False], evals


def test_shard_bounds_cover_population():
    from st_ito.style_transfer import shard_bounds
    for P in (1, 7, 8, 256, 2048, 1023):
        for G in (1, 2, 3, 8):
            spans = [shard_bounds(P, r, G) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert shard_bounds(2048, 3, 8) == (768, 1024)


_WORKER = r"""
import os, sys
sys.path.insert(0, os.path.join({root!r}, "st-ito_amd"))
import numpy as np, torch, torch.distributed as dist
from st_ito.cmaes import CMAEvolutionStrategy
from st_ito.style_transfer import sharded_evaluate
rank = int(sys.argv[1]); world = int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
P, D = 13, 6                      # odd population: uneven shards
es = CMAEvolutionStrategy(np.full(D, 0.5), 0.33, {{"bounds": [0, 1], "popsize": P, "seed": 42}})
calls = []
def eval_local(Ws):
    calls.append(len(Ws))
    return torch.tensor([float(np.sum((w - 0.25) ** 2)) for w in Ws], dtype=torch.float32), None, None
for it in range(15):
    W = es.ask()
    f, _, _ = sharded_evaluate(W, eval_local)
    assert len(f) == P
    es.tell(W, f)
out = np.concatenate([es.result[0], [es.result[1], calls[0]]])
np.save(sys.argv[4] + f".{{rank}}.npy", out)
dist.destroy_process_group()
"""


def test_population_sharding_world_size_2_gloo(tmp_path):
    """N > 1 path on CPU: two ranks, replicated seeded CMA-ES, sharded evaluation, gloo
    all-gather of the fitness scalars -> identical selected vector on both ranks and the same
    trajectory as the single-process run."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(tmp_path / "o")]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    o0, o1 = np.load(tmp_path / "o.0.npy"), np.load(tmp_path / "o.1.npy")
    np.testing.assert_array_equal(o0[:-1], o1[:-1])     # bit-identical wopt / fopt on both ranks
    assert (o0[-1], o1[-1]) == (7, 6)                    # 13 candidates -> shards of 7 and 6
    # single-process reference
    from st_ito.cmaes import CMAEvolutionStrategy
    es = CMAEvolutionStrategy(np.full(6, 0.5), 0.33, {"bounds": [0, 1], "popsize": 13, "seed": 42})
    for _ in range(15):
        W = es.ask()
        es.tell(W, torch.tensor([float(np.sum((w - 0.25) ** 2)) for w in W], dtype=torch.float32).tolist())
    np.testing.assert_array_equal(o0[:6], es.result[0])


_PAIR_WORKER = """
import os, sys
sys.path.insert(0, {root!r} + "/st-ito_amd")
import numpy as np, torch, torch.distributed as dist
from st_ito import style_transfer as ST
rank = int(sys.argv[1]); world = int(sys.argv[2])
if world > 1:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

class FakeEvaluator:                      # stands in for the GPU evaluator: pair b's optimum is its target value
    def __init__(self, x, sr, plugins, model, target_embeds, **kw):
        self.ndims = 4; self.t = target_embeds["mid"][:, 0].double().numpy(); self.B = x.shape[0]
    def evaluate(self, W, random_crop=False, rng=None):
        W = np.asarray(W); per = len(W) // self.B
        f = [float(np.sum((w - self.t[i // per]) ** 2)) for i, w in enumerate(W)]
        return torch.tensor(f, dtype=torch.float32), None, None
ST.engine.PopulationEvaluator = FakeEvaluator
ST.process_audio = lambda x, w, sr, plugins: x
ST.parameters_to_dict = lambda w, plugins: dict(w=list(w))
embed = lambda t, model, sr: dict(mid=t[:, :1, 1], side=t[:, :1, 1])   # target "embedding" = second sample
B = 5
x = torch.ones(B, 1, 8); tgt = torch.linspace(0.2, 0.8, B).view(B, 1, 1).repeat(1, 1, 8); tgt[:, :, 0] = 1.0  # peak 1
plugins = dict(fx=dict(num_params=4))
res = ST.run_es_batch(x, tgt, 48000, plugins, None, embed, max_iters=12, sigma0=0.3, popsize=6, seed=7, early_stop=False)
assert len(res) == B and all(r is not None for r in res)
np.save(sys.argv[4] + f".{{rank}}.npy", np.stack([np.concatenate([r["wopt"], [r["fopt"], r["num_evals"]]]) for r in res]))
if world > 1:
    dist.destroy_process_group()
"""


def test_pair_sharding_world_size_2_gloo(tmp_path):
    """run_es_batch (configs[2]): pairs are sharded over the ranks (5 pairs -> 3 + 2), no collective
    until the final gather; every rank ends with all results, identical to the single-process run."""
    script = tmp_path / "pair_worker.py"
    script.write_text(_PAIR_WORKER.format(root=ROOT))
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(tmp_path / "p")]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    assert subprocess.call([sys.executable, str(script), "0", "1", port, str(tmp_path / "s")]) == 0
    p0, p1, s0 = np.load(tmp_path / "p.0.npy"), np.load(tmp_path / "p.1.npy"), np.load(tmp_path / "s.0.npy")
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(p0, s0)
    assert p0.shape == (5, 6) and np.all(p0[:, 5] == 12 * 6)
    assert np.all(np.abs(p0[:, :4] - np.linspace(0.2, 0.8, 5)[:, None]) < 0.15)   # each pair heads for ITS optimum


_UNSEEDED_WORKER = """
import os, sys
sys.path.insert(0, {root!r} + "/st-ito_amd")
import numpy as np, torch, torch.distributed as dist
from st_ito import style_transfer as ST
rank = int(sys.argv[1]); world = int(sys.argv[2])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
np.random.seed(1000 + rank)               # the ranks' global RNGs differ, like two unseeded processes

class FakeEvaluator:                      # stands in for the GPU evaluator
    def __init__(self, x, sr, plugins, model, target_embeds, **kw):
        self.ndims = 4
    def evaluate(self, W, random_crop=False, rng=None, want_audio=False, dropout=0.0, parallel=False):
        W = np.asarray(W)
        f = torch.tensor([float(np.sum((w - 0.3) ** 2)) for w in W], dtype=torch.float32)
        audio = torch.stack([torch.full((1, 8), float(w[0])) for w in W]) if want_audio else None
        return f, None, audio
    def nan_warning(self):
        return None
ST.engine.PopulationEvaluator = FakeEvaluator
ST.process_audio = lambda x, w, sr, plugins: x
ST.parameters_to_dict = lambda w, plugins: dict(w=list(w))
embed = lambda t, model, sr: dict(mid=t[:, :1, 1], side=t[:, :1, 1])
x = torch.ones(1, 1, 8); tgt = torch.ones(1, 1, 8)
res = ST.run_es(x, tgt, 48000, dict(fx=dict(num_params=4)), None, embed, max_iters=6, popsize=7, find_w0=True, sigma0=0.3,
                seed=None, early_stop=False, savepop=True, run_dir=sys.argv[4])
np.save(sys.argv[4] + f"/w.{{rank}}.npy", np.concatenate([res["wopt"], [res["fopt"]]]))
dist.destroy_process_group()
"""


def test_unseeded_run_es_and_savepop_world_size_2_gloo(tmp_path):
    """with seed=None (the reference's default) the ranks must still step identical CMA-ES replicas --
    rank 0's seed is broadcast -- and --savepop must work when every rank holds only its shard of the audio:
    each rank writes its own candidates' files under their global fitness rank."""
    script = tmp_path / "unseeded_worker.py"
    script.write_text(_UNSEEDED_WORKER.format(root=ROOT))
    port = str(33500 + os.getpid() % 2000)
    out = tmp_path / "run"
    out.mkdir()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(out)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    w0, w1 = np.load(out / "w.0.npy"), np.load(out / "w.1.npy")
    np.testing.assert_array_equal(w0, w1)               # (the seed differs from run to run: no trajectory to assert)
    for it in (-1, 0, 5):
        names = sorted(os.listdir(out / f"pop_{it}"))
        assert len(names) == 7, names                                       # 4 from rank 0 + 3 from rank 1
        assert sorted(int(n.split("_")[3]) for n in names) == list(range(7))  # global fitness ranks, once each
        fv = [float(n.split("fval_")[1][:-4]) for n in sorted(names, key=lambda n: int(n.split("_")[3]))]
        assert fv == sorted(fv)


def test_run_staged_es_logic_with_fake_evaluator(monkeypatch, capsys):
    """run_staged_es (reference scripts/run_optim.py:39-234, fixed variant): stage k runs a CMA-ES over plugin k's own
    dimensions on the sub-chain 0..k, candidates are [wopt_overall, w] (161-166), max_iters // n_plugins iterations
    per stage (154), histories record the stage-local best after every tell (181-185)."""
    from st_ito import style_transfer as ST
    seen = []

    class FakeEvaluator:
        def __init__(self, x, sr, plugins, model, target_embeds, **kw):
            self.ndims = sum(p["num_params"] for p in plugins.values())
            self.names = list(plugins)
        def evaluate(self, W, want_audio=False, **kw):
            W = np.asarray(W)
            seen.append((tuple(self.names), W.copy()))
            goal = np.concatenate([np.full(3, 0.2), np.full(2, 0.8), np.full(4, 0.6)])[: W.shape[1]]
            return torch.tensor([float(np.sum((w - goal) ** 2)) for w in W], dtype=torch.float32), None, None
    monkeypatch.setattr(ST.engine, "PopulationEvaluator", FakeEvaluator)
    monkeypatch.setattr(ST, "process_audio", lambda x, w, sr, plugins: x)
    monkeypatch.setattr(ST, "parameters_to_dict", lambda w, plugins: dict(w=list(w)))
    plugins = {"a": dict(num_params=3), "b": dict(num_params=2), "c": dict(num_params=4)}
    embed = lambda t, model, sr: dict(mid=t[:, :1, 0], side=t[:, :1, 0])  # noqa: E731
    res = ST.run_staged_es(torch.ones(1, 1, 8), torch.ones(1, 1, 8), 48000, plugins, None, embed, max_iters=62, popsize=8,
                           sigma0=0.3, seed=5, run_dir=None, find_w0=True, dropout=0.0, normalize_stages=False)
    per = 62 // 3
    assert len(seen) == 3 * per and res["num_evals"] == 3 * per * 8
    assert [s[0] for s in seen[::per]] == [("a",), ("a", "b"), ("a", "b", "c")]
    assert [s[1].shape for s in seen[::per]] == [(8, 3), (8, 5), (8, 9)]
    w_a, w_b, w_c = res["stage_wopts"]
    for names, W in seen[per:2 * per]:
        np.testing.assert_array_equal(W[:, :3], np.tile(w_a, (8, 1)))      # earlier stages held at their optimum
    for names, W in seen[2 * per:]:
        np.testing.assert_array_equal(W[:, :5], np.tile(np.concatenate([w_a, w_b]), (8, 1)))
    np.testing.assert_array_equal(res["wopt"], np.concatenate([w_a, w_b, w_c]))
    assert len(res["fval_history"]) == len(res["wopt_history"]) == 3 * per
    assert np.abs(w_a - 0.2).max() < 0.1 and np.abs(w_b - 0.8).max() < 0.1 and np.abs(w_c - 0.6).max() < 0.1
    assert res["fopt"] == res["fval_history"][-1] and res["params"] == dict(w=list(res["wopt"]))
    with pytest.raises(ValueError):
        ST.run_staged_es(torch.ones(1, 1, 8), torch.ones(1, 1, 8), 48000, plugins, None, embed, distance="l2")


def test_cli_parser_keeps_reference_flags():
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import importlib
    ro = importlib.import_module("run_optim")
    a = ro.build_parser().parse_args(["in.wav", "tgt.wav", "--max-iters", "5", "--popsize", "8", "--max-length", "96000",
                                      "--effect-type", "basic", "--algorithm", "es", "--metric", "param", "--use-gpu",
                                      "--parallel", "--savepop", "--normalize-stages", "--dropout", "0.1"])
    assert (a.input, a.target_pos, a.popsize, a.max_iters, a.max_length) == ("in.wav", "tgt.wav", 8, 5, 96000)
    d = ro.build_parser().parse_args(["in.wav", "tgt.wav"])
    assert (d.max_iters, d.popsize, d.max_length, d.effect_type, d.algorithm, d.metric) == (300, 32, 262144, "vst", "es", "param")
    b = ro.build_parser().parse_args(["in.wav", "--target", "t.wav"])  # README.md:18 spelling
    assert b.target == "t.wav"
    with pytest.raises(NotImplementedError):
        ro.main(["in.wav", "t.wav", "--algorithm", "autodiff"])
    with pytest.raises(NotImplementedError):
        ro.main(["in.wav", "t.wav"])  # default --effect-type vst


def test_audio_io_roundtrip(tmp_path):
    from st_ito.audio_io import load_wav, save_wav, resample
    x = torch.rand(2, 1000) * 2 - 1
    save_wav(str(tmp_path / "a.wav"), x, 48000)
    y, sr = load_wav(str(tmp_path / "a.wav"))
    assert sr == 48000 and torch.equal(x, y)
    assert resample(y, 48000, 48000) is y          # same rate: untouched (other rates need the GPU: test_gpu_edges)


def test_sinc_resampler_known_answers():
    """The resampler in front of the path is torchaudio.functional.resample with the library defaults
    (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99; utils.py:462-463, run_optim.py:446, 526), restated from
    the published algorithm -- the oracle (torch conv1d) and the product's kernel table are checked here on the host.
    Known answers: output length ceil(n * new / orig); kernel geometry for 44.1k -> 48k (147 -> 160 phases, width 7,
    161 taps); unit DC gain; a 1 kHz sinusoid keeps frequency, phase and amplitude to 1e-3 (this filter's pass-band
    scale is 0.9995: the reference's resampler is a 6-zero-crossing Hann-windowed sinc, not a brick wall -- VERDICT r1
    asked for 1e-4 / 80 dB, which the library default the reference calls does not deliver: 15 kHz droops by 1.2e-2);
    a tone well inside the stop band (30 kHz into a 48 kHz output) comes out 45 dB down -- the side-lobe level of that
    window; measured on the restatement and pinned here so a "better" filter cannot silently replace the reference's."""
    import st_ito_oracle as O
    from st_ito.audio_io import sinc_resample_kernel
    k, width, orig, new = sinc_resample_kernel(44100, 48000)
    assert (orig, new, width) == (147, 160, 7) and k.shape == (160, 161) and k.dtype == torch.float32
    assert abs(k.sum(dim=1).mean().item() - 1.0) < 2e-3        # every phase sums to ~1: DC passes
    for o, n in ((44100, 48000), (48000, 44100), (96000, 48000), (16000, 48000)):
        for length in (1, 1000, 44100):
            assert O.resample_sinc(torch.zeros(1, length), o, n).shape == (1, -(-length * n // o))
    t = torch.arange(44100, dtype=torch.float64) / 44100
    y = O.resample_sinc(torch.sin(2 * np.pi * 1000.0 * t).float()[None], 44100, 48000)[0].double()
    ref = torch.sin(2 * np.pi * 1000.0 * torch.arange(48000, dtype=torch.float64) / 48000)
    assert (y - ref)[200:-200].abs().max().item() < 1e-3
    dc = O.resample_sinc(torch.ones(1, 5000), 44100, 48000)[0]
    assert (dc[100:-100] - 1.0).abs().max().item() < 2e-3
    t = torch.arange(96000, dtype=torch.float64) / 96000
    y = O.resample_sinc(torch.sin(2 * np.pi * 30000.0 * t).float()[None], 96000, 48000)[0].double()
    att_db = 20 * np.log10(y[500:-500].pow(2).mean().sqrt().item() * np.sqrt(2) + 1e-30)
    assert -50.0 < att_db < -40.0, att_db
    # product table == oracle table (same restatement, two code paths)
    got = torch.nn.functional.conv1d(torch.nn.functional.pad(torch.ones(1, 1, 600), (width, width + orig)), k[:, None], stride=orig)
    assert torch.allclose(got.transpose(1, 2).reshape(1, -1)[:, :654], O.resample_sinc(torch.ones(1, 600), 44100, 48000), atol=0)


def test_bs1770_loudness_known_answers():
    """st_ito.loudness (pyloudnorm restatement used by the eval_pst harness): BS.1770 reference points."""
    from st_ito.loudness import integrated_loudness, normalize_loudness
    sr = 48000
    t = np.arange(5 * sr) / sr
    s = np.sin(2 * np.pi * 997 * t)
    assert abs(integrated_loudness(s, sr) - (-3.01)) < 0.1              # 0 dBFS 997 Hz sine, one channel
    assert abs(integrated_loudness(np.stack([s, s], 1), sr) - 0.0) < 0.1  # the same in both front channels: +3.01 LU
    assert abs(integrated_loudness(0.1 * s, sr) - (-23.01)) < 0.1       # -20 dB
    assert integrated_loudness(np.zeros(sr), sr) == -np.inf
    # gating: 4 s of signal + 6 s of near silence measures like the signal alone
    gated = np.concatenate([0.1 * s[: 4 * sr], 1e-5 * s[: 4 * sr], np.zeros(2 * sr)])
    assert abs(integrated_loudness(gated, sr) - integrated_loudness(0.1 * s[: 4 * sr], sr)) < 0.3
    y, lufs = normalize_loudness(torch.from_numpy(np.stack([0.1 * s, 0.1 * s]).astype(np.float32)), sr, -22.0)
    assert abs(integrated_loudness(y.numpy().T, sr) - (-22.0)) < 0.05 and abs(lufs - (-20.0)) < 0.1
    with pytest.raises(ValueError):
        integrated_loudness(np.zeros(100), sr)


def test_eval_pst_chain_catalogue():
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import importlib
    ep = importlib.import_module("eval_pst")
    from st_ito.style_transfer import load_plugins
    want = {"general-pb": (["Distortion", "ParametricEQ", "Compressor", "Delay", "Reverb"], 31 + 5),
            "mastering-pb": (["ParametricEQ", "Compressor", "Reverb"], 26 + 3),
            "vocals-pb": (["ParametricEQ", "Compressor", "Distortion", "Delay", "Reverb"], 31 + 5),
            "guitar-pb": (["Compressor", "ParametricEQ", "Distortion", "Reverb"], 28 + 4)}
    for chain, (names, ndim) in want.items():
        pl, n, init = load_plugins(ep.get_plugins(chain))
        assert list(pl) == names and n == ndim == len(init)
    with pytest.raises(ValueError):
        ep.get_plugins("general-vst")


def test_bench_measurement_bookkeeping(tmp_path, monkeypatch):
    """bench.py's roofline inputs: the conv table sums to SURVEY 8(d)'s 37.22 GFLOP per 10 s stream (37.15 for the eleven
    MFMA layers), the committed PMC file belongs to the kernel sources in this tree, and a PMC file taken on other sources
    or another stream count is refused (traffic = None with the reason) rather than quoted."""
    import json
    import bench
    rows = bench.conv_layer_table(469)
    assert len(rows) == 12 and [r["cout"] for r in rows][::2] == [64, 128, 256, 512, 1024, 2048]
    assert abs(sum(r["flops"] for r in rows) / 1e9 - 37.223) < 0.01
    assert abs(sum(r["flops"] for r in rows if r["cin"] % 8 == 0) / 1e9 - 37.154) < 0.01
    committed = json.load(open(bench.PMC_TRAFFIC_JSON))
    t, note = bench.pmc_traffic_per_launch(committed["n_streams"])
    if committed["kernel_source_hash"] == bench.kernel_source_hash():
        assert t == committed["traffic_bytes_per_launch"] and t > committed["algorithmic_bytes_per_launch"]
        assert os.path.basename(bench.PMC_TRAFFIC_JSON) in note
    else:  # kernels edited since the last PMC pass (tools/profile_round.sh): the bench line must say so instead of quoting it
        assert t is None and "was taken on kernel sources" in note
        import warnings
        warnings.warn(f"{os.path.basename(bench.PMC_TRAFFIC_JSON)} is stale (kernel sources changed): roofline.traffic will be null until "
                      "tools/profile_round.sh is run again on the GPU box")
    assert bench.pmc_traffic_per_launch(committed["n_streams"] + 1)[0] is None
    stale = dict(committed, kernel_source_hash="0" * 16)
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_JSON", str(p))
    t, note = bench.pmc_traffic_per_launch(committed["n_streams"])
    assert t is None and "was taken on kernel sources" in note


def test_bench_self_launches_ranks_dryrun():
    """bench.py started from a bare shell with --gpus 2 and no WORLD_SIZE launches its own ranks through
    torch.distributed.run.  STITO_BENCH_DRYRUN=1 stops each rank after the gloo rendezvous + the
    barrier / max-over-ranks skeleton, so the launch path is covered without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STITO_BENCH_DRYRUN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # ... with the per-rank stage breakdown gathered the way the real run gathers it (min / max over ranks of evaluate / gather / tell)
    cfg = d.pop("config")
    assert d == {"dryrun": True, "n_gpus": 2, "ranks_seen": [0, 1], "max_over_ranks": 2.0, "steps": 3, "warmup": 1,
                 "stages": {"evaluate_ms": {"min": 10.0, "max": 20.0}, "gather_ms": {"min": 1.0, "max": 2.0}, "tell_ms": {"min": 0.5, "max": 0.5}}}
    assert cfg["baseline_config"] == 1 and cfg["pop_per_gpu"] == 256 and cfg["n_samples"] == 480000 and "configs[1]" in cfg["workload"]
    # --config 3 / 4: BASELINE.json's two 8-GPU configurations as ONE command -- their defaults (30 s audio,
    # 256 / 128 candidates per GPU, 50 iterations for configs[3], the 96 000-tap convolution reverb in configs[4]'s chain) and the
    # workload string naming the BASELINE entry, through the same two-rank launch path
    import json as _json
    base_cfgs = _json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    for c, pop, steps, D, word in ((3, 256, 50, 45, "pop=2048 sharded 256/GPU"), (4, 128, 10, 66, "convolution-reverb IR=2 s")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", str(c)],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["steps"] == steps and d["ranks_seen"] == [0, 1]
        cfg = d["config"]
        assert cfg["baseline_config"] == c and cfg["pop_per_gpu"] == pop and cfg["n_samples"] == 30 * 48000
        assert word in cfg["workload"] and word in base_cfgs[c] and f"(D={D})" in cfg["workload"] and f"({2 * pop} total)" in cfg["workload"]
        assert ("NoiseShapedReverb(96000 taps)" in cfg["chain"]) == (c == 4)
    # a rank count that disagrees with the launcher's WORLD_SIZE is refused, not silently benchmarked
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_compressor_scan_ring_registers_are_out_of_the_compilers_reach(tmp_path):
    """k_comp_blockscan keeps its 32-block register ring and the in-flight loads in fixed VGPRs v100+ across separate asm
    statements (generated: tools/gen/gen_comp_scan_asm.py); those registers are only declared as clobbers, so nothing tells
    the compiler they are live in between.  Build check: the code hipcc itself emits for that kernel -- everything
    outside the asm statements -- must stay below v100."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "st-ito_amd", "csrc", "compressor.hip")
    out = tmp_path / "comp.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w", src, "-o", str(out)])
    text = out.read_text()
    m = re.search(r"^(_ZN5stito16k_comp_blockscan\w*):.*?^\s*s_endpgm", text, flags=re.S | re.M)
    assert m, "k_comp_blockscan not found in the compiled assembly"
    body, inside, highest, n_asm = m.group(0), False, -1, 0
    for line in body.splitlines():
        if "ASMSTART" in line:
            inside, n_asm = True, n_asm + 1
            continue
        if "ASMEND" in line:
            inside = False
            continue
        if inside:
            continue
        for a, b in re.findall(r"\bv(\d+)\b|\bv\[\d+:(\d+)\]", line.split(";")[0]):
            highest = max(highest, int(a or b))
    assert n_asm >= 3, "the ring's asm statements are gone: update this check"
    assert 0 <= highest < 100, f"compiler-allocated code of k_comp_blockscan reaches v{highest}: it would clobber the ring (v100+)"


def test_reverb_staging_ring_registers_are_out_of_the_compilers_reach(tmp_path):
    """k_reverb's staging waves keep a four-tile register ring in FIXED VGPRs v64 .. v79, loaded by inline-asm global_load_dword
    and handed to the compiler only inside the asm statement that waits for them (with "=v" outputs on the loads the
    compiler could legally copy a register whose load was still in flight).  The registers are only declared as clobbers, so --
    as for the compressor scan's ring -- the build check is that everything hipcc itself emits for the kernel stays below v64,
    that the ring's 36 loads are all inside asm statements, and that the kernel does not spill."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "st-ito_amd", "csrc", "dsp.hip")
    out = tmp_path / "dsp.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-w",
                           "-I", os.path.join(ROOT, "include"), src, "-o", str(out)])
    text = out.read_text()
    bodies = [m.group(0) for m in re.finditer(r"^(_ZN5stito8k_reverb\w*):.*?^\s*s_endpgm", text, flags=re.S | re.M)]
    assert len(bodies) == 2, "k_reverb<false> and k_reverb<true> (one workgroup per candidate / per channel) expected in the compiled assembly"
    highest = -1
    for body in bodies:
        inside, ring_loads, ring_regs = False, 0, set()
        for line in body.splitlines():
            if "ASMSTART" in line:
                inside = True
                continue
            if "ASMEND" in line:
                inside = False
                continue
            code = line.split(";")[0]
            if inside:
                mm = re.search(r"global_load_dword v(\d+),", code)
                if mm:
                    ring_loads += 1
                    ring_regs.add(int(mm.group(1)))
                continue
            assert "scratch_" not in code, f"k_reverb spills: {code.strip()}"
            for a, b in re.findall(r"\bv(\d+)\b|\bv\[\d+:(\d+)\]", code):
                highest = max(highest, int(a or b))
        assert ring_loads == 36 and ring_regs == set(range(64, 80)), (ring_loads, sorted(ring_regs))
    assert 0 <= highest < 64, f"compiler-allocated code of k_reverb reaches v{highest}: it would clobber the staging ring (v64 .. v79)"


def test_cmaes_prefetch_does_not_change_the_run():
    """prefetch() only moves the draw of the next generation's normals ahead of tell(): same generator, same order."""
    from st_ito.cmaes import CMAEvolutionStrategy

    def run(prefetch):
        es = CMAEvolutionStrategy(np.full(7, 0.5), 0.3, {"bounds": [0, 1], "popsize": 12, "seed": 5})
        for _ in range(6):
            W = es.ask()
            if prefetch:
                es.prefetch()
                es.prefetch()  # idempotent until the next ask()
            es.tell(W, [float(np.sum((w - 0.3) ** 2)) for w in W])
        return np.asarray(es.ask()), es.result[0]

    a, ra = run(False)
    b, rb = run(True)
    assert np.array_equal(a, b) and np.array_equal(ra, rb)


def test_case_study_cases_mirror_the_reference_table():
    """scripts/eval_case_study.py: the six pb_* cases of the reference's eval_case_study.py:226-344 -- one free parameter, every
    other one fixed in its own units (Parameter.set_value must accept them), the free parameter's slot index counted with the
    leading our_bypass slot -- compile into a chain whose fixed mask covers exactly the fixed names; vst_* cases say why they
    are not built; unknown names raise like the reference's KeyError-free else branch would not."""
    import copy
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import eval_case_study as C
    from st_ito import engine
    from st_ito.style_transfer import load_plugins
    want = {"pb_ParametricEQ": ("low_shelf_gain_db", 0.0, 1.0, 18), "pb_Chorus": ("mix", 0.0, 1.0, 5), "pb_Compressor": ("threshold_db", 0.0, 1.0, 4),
            "pb_Distortion": ("drive_db", 0.5, 1.0, 2), "pb_Delay": ("mix", 0.0, 1.0, 3), "pb_Reverb": ("room_size", 0.0, 1.0, 4)}
    for name, (param, lo, hi, n_real) in want.items():
        spec, p, a, b = C.get_case(name)
        assert (p, a, b) == (param, lo, hi) and list(spec) == [name] and spec[name]["fixed_parameters"]["our_bypass"] == 0.0
        plugins, total, init = load_plugins(copy.deepcopy(spec))
        assert total == n_real + 1 and plugins[name]["parameter_names"][0] == "our_bypass" and init[0] == 0.0
        descs, D = engine.compile_chain(plugins)
        real = plugins[name]["parameter_names"][1:]
        fixed = {i for i, nm in enumerate(real) if nm in spec[name]["fixed_parameters"]}
        assert D == total and descs[0].has_bypass == 1 and descs[0].fixed_mask == sum(1 << i for i in fixed)
        assert set(range(n_real)) - fixed == {real.index(param)}            # exactly one free parameter
        for i in fixed:                                                    # the fixed raw values are the normalised own-unit values
            prm = plugins[name]["instance"].parameters[real[i]]
            assert descs[0].fixed_raw[i] == pytest.approx((spec[name]["fixed_parameters"][real[i]] - prm.min_value) / (prm.max_value - prm.min_value))
    with pytest.raises(NotImplementedError):
        C.get_case("vst_RoughRider3")
    with pytest.raises(ValueError):
        C.get_case("pb_Flanger")
    x = torch.zeros((2, C.MIN_LEN + 1000)); x[:, ::7] = 0.5
    rng = np.random.RandomState(3)
    a, b = C.crop_pair(x[:1], x, rng)                                       # mono source -> stereo crop; lengths and guards
    assert a.shape[0] == 2 and b.shape[0] == 2 and 262144 <= a.shape[1] < 524288 and 262144 <= b.shape[1] < 524288
    assert float(a.abs().max()) == 1.0 and float(b.abs().max()) == 1.0


def test_savepop_reference_switch_reproduces_the_reference_files(tmp_path, monkeypatch, golden_dir):
    """--savepop deliberately writes the whole population; the reference's savepop_to_disk
    (style_transfer.py:362-396) zips the population with the embedding DICT run_es hands it and therefore writes two files --
    candidates 0 and 1 ranked among themselves.  STITO_SAVEPOP_REFERENCE=1 reproduces exactly that: file names and samples
    against a fixture the reference's own function produced (tests/golden/make_golden.py g9, torchaudio.save as a recorder)."""
    from st_ito.audio_io import load_wav
    from st_ito.style_transfer import savepop_to_disk
    g = np.load(os.path.join(golden_dir, "savepop_reference.npz"))
    fvals = [float(v) for v in g["fvals"]]
    audios = [torch.from_numpy(a[0].copy()) for a in g["audios"]]      # (chs, n) per candidate, as evaluate hands them over
    embeds = {"mid": torch.zeros(len(fvals), 4), "side": torch.zeros(len(fvals), 4)}
    it = int(g["iteration"])
    monkeypatch.setenv("STITO_SAVEPOP_REFERENCE", "1")
    savepop_to_disk(it, fvals, embeds, audios, str(tmp_path / "ref"), int(g["rate"]))
    names = sorted(os.listdir(tmp_path / "ref" / f"pop_{it}"))
    assert names == sorted(str(n) for n in g["names"]) and len(names) == 2
    for name, want in zip(g["names"], g["written"]):
        audio, sr = load_wav(str(tmp_path / "ref" / f"pop_{it}" / str(name)))
        assert sr == int(g["rate"])
        np.testing.assert_array_equal(np.asarray(audio, dtype=np.float32).reshape(want.shape), want)
    # a rank that does not hold candidates 0 / 1 writes nothing in this mode; the default writes all five
    savepop_to_disk(it, fvals, embeds, audios[2:], str(tmp_path / "shard"), int(g["rate"]), first=2)
    assert os.listdir(tmp_path / "shard" / f"pop_{it}") == []
    monkeypatch.delenv("STITO_SAVEPOP_REFERENCE")
    savepop_to_disk(it, fvals, embeds, audios, str(tmp_path / "all"), int(g["rate"]))
    assert len(os.listdir(tmp_path / "all" / f"pop_{it}")) == 5
