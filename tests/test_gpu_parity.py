"""GPU parity tests: the HIP path (through the C ABI, via st_ito's ctypes binding) against the
CPU oracle on the same seeded inputs and against the golden vectors generated from the
reference.  Tolerances: bit-level for index/shape logic, float tolerances stated per test
(north_star: embeddings and losses within 1e-4 relative fp32).

Run with:  python -m pytest tests -m gpu
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import st_ito_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito import _hip
    _hip.lib()  # must load: no silent fallback
    return torch.device("cuda", 0)


def _plugins_pair(kinds, with_bypass=False):
    """Same chain as oracle plugins and as product plugins."""
    from st_ito import effects as E
    cls = {"ParametricEQ": (E.BasicParametricEQ, 1), "Compressor": (E.BasicCompressor, 1),
           "Distortion": (E.BasicDistortion, 1), "Delay": (E.BasicDelay, 2), "Reverb": (E.BasicReverb, 2),
           "Gain": (E.BasicGain, 1)}
    spec, seen = [], {}
    for k in kinds:
        seen[k] = seen.get(k, 0) + 1
        name = k if seen[k] == 1 else f"{k}{seen[k]}"
        spec.append((name, cls[k][0], cls[k][1]))
    return O.make_plugins(kinds, with_bypass), E.make_plugins(spec, with_bypass)


def _render_gpu(pp, x, W, dev, normalize=True):
    from st_ito import engine
    audio, peaks = engine.render_population(pp, torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), SR)
    if normalize:
        engine.normalize_audio_(audio, peaks)
    return audio.cpu().numpy(), peaks.cpu().numpy()


def _oracle_chain_raw(op, x, w):
    """Oracle chain output WITHOUT the final peak normalisation."""
    widx, y = 0, x
    for name, plugin in op.items():
        for pn in plugin["parameter_names"]:
            if pn != "our_bypass":
                plugin["instance"].parameters[pn].raw_value = w[widx]
            widx += 1
        if plugin["num_channels"] == 2 and y.shape[0] == 1:
            y = np.concatenate((y, y), 0)
        if plugin["num_channels"] == 1 and y.shape[0] == 2:
            y = np.concatenate((plugin["instance"].process(y[0:1], SR), plugin["instance"].process(y[1:2], SR)), 0)
        else:
            y = plugin["instance"].process(y, SR)
    return y


SINGLE_FX = [
    ("ParametricEQ", 1, 2e-6), ("ParametricEQ", 2, 2e-6), ("Compressor", 2, 5e-6), ("Distortion", 1, 2e-6),
    ("Gain", 2, 1e-6), ("Delay", 1, 2e-6), ("Delay", 2, 2e-6), ("Reverb", 2, 2e-5), ("Reverb", 1, 2e-5),
]


@pytest.mark.parametrize("kind,chs,tol", SINGLE_FX)
def test_single_effect_vs_oracle(dev, kind, chs, tol):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"{kind}{chs}".encode()) % 1000)  # (hash() of a str changes from process to process)
    n, P = 30011, 5
    x = O.synth_audio(5, chs, n).numpy()
    op, pp = _plugins_pair([kind])
    D = sum(p["num_params"] for p in op.values())
    W = rng.random((P, D))
    W[0] = 0.0; W[1] = 1.0  # range extremes
    got, _ = _render_gpu(pp, x, W, dev, normalize=False)
    for p in range(P):
        ref = _oracle_chain_raw(op, x, W[p])
        assert got[p].shape == ref.shape
        scale = max(1.0, np.abs(ref).max())
        err = np.abs(got[p] - ref).max() / scale
        print(f"{kind} {chs}ch cand {p}: max rel-to-peak err {err:.3e}")
        assert err < tol, f"{kind} cand {p}: max err {err:.3e}"


def test_compressor_ballistics_corners(dev):
    """Heavy compression (threshold -40 dB, ratio 20) at the corners of the attack/release range,
    including attack slower than release (c_att > c_rel: the GPU envelope runs on z = -y there) and
    an odd length.  The GPU evaluates the switching one-pole as max(c_a y + (1-c_a) v, c_r y + (1-c_r) v)
    and crosses 15-sample blocks with their composed (max, +) block functions (compressor.hip); the oracle
    walks v + c (y - v) sample by sample: the same real-number map with different float32 rounding
    (measured here: <= 2.4e-7 of peak; bound 2e-6 like the other compressor test)."""
    op, pp = _plugins_pair(["Compressor"])
    raw = lambda v, lo, hi: (v - lo) / (hi - lo)
    rows = []
    for att in (0.1, 1.0, 100.0):
        for rel in (10.0, 1000.0):
            rows.append([raw(-40.0, -80.0, 0.0), raw(20.0, 1.0, 20.0), raw(att, 0.1, 100.0), raw(rel, 10.0, 1000.0)])
    W = np.array(rows)
    for n in (48000, 30011):
        x = O.synth_audio(9, 2, n).numpy()
        got, _ = _render_gpu(pp, x, W, dev, normalize=False)
        for p in range(len(W)):
            ref = _oracle_chain_raw(op, x, W[p])
            assert np.abs(ref).max() < 0.5 * np.abs(x).max()  # it really compresses
            err = np.abs(got[p] - ref).max() / max(1.0, np.abs(ref).max())
            print(f"compressor att/rel corner {p} n={n}: err {err:.3e}")
            assert err < 2e-6


@pytest.mark.parametrize("n", [1, 14, 15, 16, 29, 30, 31, 479, 480, 481, 495, 7681])
def test_compressor_block_boundaries(dev, n):
    """Lengths around the compressor's block structure: shorter than one 15-sample block (no block function at
    all), exact multiples, one sample either side, one register ring of 32 blocks (480 samples) +- 1, one past
    a 256-block apply tile + ring; 3 candidates x 1 channel = 3 streams (a partly idle quad group) and
    9 x 2 = 18 streams (two serial waves, the second nearly idle).

    Attack times are kept <= 10 ms here.  With a slow attack the first samples of the reference's own recurrence,
    v + c (y - v) with c within 1e-3 of 1 and y << v, carry rounding noise of ulp(v) / y ~ 1e-5 relative, which
    the gain computer passes on (measured 3.6e-5 of peak at n = 16, attack 67 ms): there the reference's output is
    its own float32 noise and no other order of operations reproduces it.  The GPU walks the first block in the
    reference's order (bit-identical) and starts every later block from the (max, +) boundary state;
    test_compressor_ballistics_corners covers the slow-attack corners on full-length audio."""
    op, pp = _plugins_pair(["Compressor"])
    rng = np.random.default_rng(n)
    for P, chs in ((3, 1), (9, 2)):
        x = (0.7 * rng.standard_normal((chs, n))).astype(np.float32)
        W = rng.random((P, 4))
        W[:, 0] = rng.uniform(0.3, 0.6, P)  # thresholds -56 .. -32 dB: always compressing
        W[:, 2] = rng.uniform(0.0, 0.099, P)  # attack 0.1 .. 10 ms
        got, _ = _render_gpu(pp, x, W, dev, normalize=False)
        for p in range(P):
            ref = _oracle_chain_raw(op, x, W[p])
            err = np.abs(got[p] - ref).max() / max(1.0, np.abs(ref).max())
            assert got[p].shape == ref.shape and err < 2e-5, f"n={n} P={P} cand {p}: {err:.3e}"


def _noise_reverb_pair(n_taps, taps=255):
    """Oracle and product NoiseShapedReverb plugin dicts sharing one seeded noise bank."""
    from st_ito import effects as E
    bank = O.make_noise_bank(n_taps, taps, SR, seed=3)
    assert torch.equal(bank, E.make_noise_bank(n_taps, taps, SR, seed=3))  # the two restatements agree bitwise
    oi, pi = O.OracleNoiseShapedReverb(noise_bank=bank), E.NoiseShapedReverb(noise_bank=bank)
    names = list(oi.parameters.keys())
    assert names == list(pi.parameters.keys()) and len(names) == 25
    mk = lambda inst: {"ConvReverb": {"class_path": type(inst), "num_params": 25, "num_channels": 2, "fixed_parameters": {},
                                       "instance": inst, "parameter_names": list(names)}}
    return mk(oi), mk(pi)


@pytest.mark.parametrize("n_taps,n,chs", [(5000, 30011, 2), (5000, 9000, 1), (65536, 20000, 2), (4096, 4096, 2)])
def test_noise_shaped_conv_reverb_vs_oracle(dev, n_taps, n, chs):
    """SURVEY 8(a) a9 / configs[4]: partitioned overlap-save FFT convolution on the GPU against the
    oracle's restatement of dasp noise_shaped_reverberation (direct conv1d for short IRs, float64 FFT
    for long ones).  float32 FFT round-off: bound 1e-5 of the output peak."""
    op, pp = _noise_reverb_pair(n_taps)
    rng = np.random.default_rng(n_taps + n)
    x = O.synth_audio(17, chs, n).numpy()
    W = rng.random((4, 25))
    W[0, 24] = 1.0   # fully wet
    W[1, :12] = 0.0  # all band gains zero: wet = 0 -> y = (1 - mix) x
    got, _ = _render_gpu(pp, x, W, dev, normalize=False)
    assert got.shape == (4, 2, n)  # always stereo
    for p in range(4):
        ref = _oracle_chain_raw(op, x, W[p])
        err = np.abs(got[p] - ref).max() / max(1e-3, np.abs(ref).max())
        print(f"conv reverb taps={n_taps} n={n} cand {p}: rel-to-peak err {err:.3e}")
        assert err < 1e-5
    xs = np.concatenate((x, x), 0) if chs == 1 else x
    np.testing.assert_allclose(got[1], (1.0 - np.float32(W[1, 24])) * xs, rtol=0, atol=1e-7)


def test_conv_reverb_in_chain_and_multi_pair(dev):
    """The stage after an in-place effect (per-candidate input blocks) and as first effect of a
    multi-pair batch (input spectra shared by each pair's candidates)."""
    from st_ito import effects as E, engine
    bank = O.make_noise_bank(6000, 255, SR, seed=5)
    def chain(mod, eq, rv, gn):
        pl = {}
        for name, inst, nch in (("ParametricEQ", eq(), 1), ("ConvReverb", rv(noise_bank=bank), 2), ("Gain", gn(), 1)):
            names = list(inst.parameters.keys())
            pl[name] = {"class_path": type(inst), "num_params": len(names), "num_channels": nch, "fixed_parameters": {},
                        "instance": inst, "parameter_names": names}
        return pl
    op = chain(O, O.OracleParametricEQ, O.OracleNoiseShapedReverb, O.OracleGain)
    pp = chain(E, E.BasicParametricEQ, E.NoiseShapedReverb, E.BasicGain)
    x = O.synth_audio(23, 1, 25000).numpy()  # mono: EQ runs mono, the reverb up-mixes
    W = np.random.default_rng(8).random((3, 18 + 25 + 1))
    got, _ = _render_gpu(pp, x, W, dev, normalize=False)
    for p in range(3):
        ref = _oracle_chain_raw(op, x, W[p])
        assert np.abs(got[p] - ref).max() / max(1e-3, np.abs(ref).max()) < 2e-5
    # reverb first, two pairs x two candidates
    _, rp = _noise_reverb_pair(5000)
    xs = torch.stack([O.synth_audio(60 + b, 2, 20000) for b in range(2)]).to(dev)
    Wm = torch.from_numpy(np.random.default_rng(9).random((4, 25))).to(dev)
    multi, _ = engine.render_population(rp, xs, Wm, SR)
    for b in range(2):
        single, _ = engine.render_population(rp, xs[b], Wm[2 * b:2 * b + 2], SR)
        assert torch.equal(multi[2 * b:2 * b + 2], single)


def test_eq_golden_reference_vectors(dev, golden_dir):
    """HIP EQ against outputs of the reference's own parametric_eq (tests/golden/eq_parametric.npz)."""
    from st_ito import effects as E
    g = np.load(os.path.join(golden_dir, "eq_parametric.npz"))
    inst = E.BasicParametricEQ()
    names = list(inst.parameters.keys())
    lo = np.array([inst.parameters[k].min_value for k in names])
    hi = np.array([inst.parameters[k].max_value for k in names])
    W = (g["params"] - lo) / (hi - lo)
    pp = E.make_plugins("eq")
    for sig, key in ((g["noise"], "y_noise"), (None, "y_impulse")):
        if sig is None:
            sig = np.zeros_like(g["noise"]); sig[0, 0] = 1.0
        got, _ = _render_gpu(pp, sig, W, dev, normalize=False)
        ref = g[key]
        for p in range(len(W)):
            scale = max(1.0, np.abs(ref[p]).max())
            assert np.abs(got[p] - ref[p]).max() / scale < 2e-6


def test_process_audio_golden(dev, golden_dir):
    """process_audio channel rules, dead bypass dimension, fixed parameters, joint peak norm
    against the reference's own process_audio (tests/golden/process_audio.npz)."""
    from st_ito import effects as E
    from st_ito.style_transfer import process_audio, parameters_to_dict
    g = np.load(os.path.join(golden_dir, "process_audio.npz"))
    for ci, (nplug, bypass, chs) in enumerate(g["cases"]):
        spec = [("ParametricEQ" if i == 0 else f"ParametricEQ{i + 1}", E.BasicParametricEQ, 1) for i in range(int(nplug))]
        pp = E.make_plugins(spec, bool(bypass))
        y = process_audio(g[f"x{ci}"].copy(), g[f"w{ci}"], SR, pp)
        np.testing.assert_allclose(y, g[f"y{ci}"], rtol=0, atol=3e-6)
        d = parameters_to_dict(g[f"w{ci}"], pp)
        flat = np.array([v for pn in d for v in d[pn].values()])
        np.testing.assert_allclose(flat, g[f"d{ci}"], rtol=1e-15)
    pp = E.make_plugins("eq")
    pp["ParametricEQ"]["fixed_parameters"] = {"band1_gain_db": 12.0, "band1_cutoff_freq": 2500.0}
    y = process_audio(g["xf"].copy(), g["wf"], SR, pp)
    np.testing.assert_allclose(y, g["yf"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("chain,chs", [
    (["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"], 2),
    (["ParametricEQ", "Compressor", "Distortion", "Delay", "Reverb"], 1),
    (["ParametricEQ", "Compressor"], 1),
])
def test_chain_vs_oracle(dev, chain, chs):
    rng = np.random.default_rng(3)
    n, P = 48000, 4
    x = O.synth_audio(9, chs, n).numpy()
    op, pp = _plugins_pair(chain, with_bypass=(chs == 1))
    D = sum(p["num_params"] for p in op.values())
    W = rng.random((P, D))
    got, peaks = _render_gpu(pp, x, W, dev)
    for p in range(P):
        ref = O.process_audio(x.copy(), W[p], SR, op)
        assert got[p].shape == ref.shape
        err = np.abs(got[p] - ref).max()
        print(f"chain {chs}ch cand {p}: max abs err {err:.3e}")
        # five cascaded float32 effects: rounding noise of an early stage is amplified by later EQ boosts (up to +24 dB per band)
        # in the oracle and in the HIP path alike; measured 5e-7 ... 2.5e-6 on these twelve renders, bound = the single-effect bar
        # (round 4 had 3e-4 here, looser than the 1e-4 north_star allows the embeddings: VERDICT r4 weak #4)
        assert err < 2e-5, f"cand {p}: {err:.3e}"
        assert abs(np.abs(got[p]).max() - 1.0) < 1e-6


def _models(dev, norm="minmax", seed=0):
    from st_ito.models.panns import Cnn14
    om = O.make_synthetic_model(seed, input_norm=norm)
    pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, norm)
    pm.load_state_dict(om.state_dict())
    pm.eval().to(dev)
    return om, pm


@pytest.mark.parametrize("norm", ["minmax", "batchnorm", "none"])
def test_logmel_vs_oracle_and_golden(dev, golden_dir, norm):
    g = np.load(os.path.join(golden_dir, f"cnn14_trunk_{norm}.npz"))
    om, pm = _models(dev, norm)
    x = torch.from_numpy(g["x"]).to(dev)
    lm = pm.logmel(x).cpu().numpy().reshape(g["logmel"].shape)
    # log-mel values live on a ~[-100, 40] dB scale (minmax: [-1, 1]); the oracle evaluates the
    # DFT as a float32 matrix product, the HIP path as a float32 FFT: agreement to ~1e-4 dB
    tol = 2e-5 if norm == "minmax" else 2e-3
    assert np.abs(lm - g["logmel"]).max() < tol
    xm = torch.from_numpy(g["x_mono"]).to(dev)
    lmm = pm.logmel(xm).cpu().numpy()
    with torch.no_grad():
        ref = om.logmel(torch.from_numpy(g["x_mono"])).numpy().reshape(lmm.shape)
    assert np.abs(lmm - ref).max() < tol


@pytest.mark.parametrize("n,chan", [(480000, 2), (262144, 2), (48001, 2), (30001, 1), (2048, 2)])
def test_logmel_wave_kernel_vs_general_kernel(dev, n, chan, monkeypatch):
    """The AFx-Rep front end runs k_logmel_wave (one wave per frame, FFT in registers, hops shared between frames); the
    general kernel (any n_fft, one workgroup per frame) stays behind STITO_LOGMEL_GENERIC=1.  Same windowed samples, same
    unpack and mel sums, different butterfly order: equal to float32 FFT rounding (1e-5 of the [-1, 1] scale except
    where the power is at the clamp), on lengths that are / are not multiples of the hop, odd lengths (unaligned right
    channel: the scalar load path), mono, and the shortest input (every hop reflected)."""
    om, pm = _models(dev, "minmax")
    x = torch.stack([O.synth_audio(11 + i, chan, n) * (0.9 if i == 0 else 0.05) for i in range(3)]).to(dev)
    monkeypatch.setenv("STITO_LOGMEL_GENERIC", "1")
    ref = pm.logmel(x).cpu().numpy()
    monkeypatch.setenv("STITO_LOGMEL_GENERIC", "0")
    got = pm.logmel(x).cpu().numpy()
    assert got.shape == ref.shape == (3 * chan, n // 1024 + 1, 128) and np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("n_fft,hop,mels", [(1024, 512, 64), (512, 128, 40), (4096, 2048, 128)])
def test_logmel_other_window_sizes(dev, n_fft, hop, mels):
    """Front ends other than the AFx-Rep one: log2(n_fft/2) odd (a radix-2 stage in front of the radix-4 ones)
    and even, different hop / mel counts -- against the oracle's torchlibrosa restatement."""
    from st_ito.models.panns import Cnn14
    om = O.fill_deterministic(O.Cnn14(512, SR, n_fft, hop, mels, 20, 20000, True, "none"), 0).eval()
    pm = Cnn14(512, SR, n_fft, hop, mels, 20, 20000, True, "none")
    pm.load_state_dict(om.state_dict())
    pm.eval().to(dev)
    x = torch.stack([O.synth_audio(95, 2, 40000), 0.2 * O.synth_audio(96, 2, 40000)])
    got = pm.logmel(x.to(dev)).cpu().numpy()
    with torch.no_grad():
        ref = om.logmel(x).numpy().reshape(got.shape)
    assert got.shape == (4, 40000 // hop + 1, mels)
    assert np.abs(got - ref).max() < 2e-3   # dB scale, like the "none" case above


CONV_CASES = [  # (n, H, W, cin, cout, pool)
    (3, 29, 8, 64, 64, 0), (2, 117, 32, 64, 64, 0), (5, 7, 2, 64, 64, 0), (2, 13, 6, 8, 64, 1),
    (2, 33, 128, 1, 64, 0), (2, 33, 128, 64, 64, 1), (3, 16, 64, 64, 128, 0), (2, 17, 32, 128, 128, 1),
    (2, 9, 16, 256, 256, 1), (3, 4, 8, 128, 256, 0), (5, 2, 4, 256, 128, 0), (3, 14, 4, 64, 128, 0), (1, 7, 4, 64, 64, 0),
    (2, 29, 8, 64, 128, 1),
    # wide layers (no per-layer case had cin > 256) and shapes that straddle streams / partial tiles
    (2, 14, 4, 512, 512, 0), (1, 14, 4, 1024, 2048, 0), (1, 14, 4, 2048, 2048, 0), (2, 29, 8, 512, 1024, 0),
    (3, 58, 16, 256, 512, 1), (2, 234, 64, 64, 128, 1), (2, 30, 10, 64, 64, 1), (4, 5, 5, 64, 64, 0), (7, 14, 4, 64, 64, 0),
    # maps the direct split-precision kernel (algo 7) tiles: 32- and 16-wide tiles, odd heights under pooling (a skipped input
    # row at every stream boundary), partial column tiles, tiles that span several short streams, 16-channel inputs
    (3, 21, 32, 64, 64, 1), (2, 117, 32, 128, 256, 0), (9, 5, 16, 16, 64, 0), (2, 35, 48, 32, 128, 1), (3, 58, 16, 256, 256, 0),
    # the register-resident F(2x2,3x3) kernel (algo 8): the bench maps of conv_block1.conv2 / conv_block2.conv1, widths that are not
    # multiples of its 32-pixel groups, odd sizes with and without pooling, one-pixel-wide and one-row maps
    (1, 469, 128, 64, 64, 1), (2, 234, 64, 64, 128, 0), (2, 37, 50, 64, 64, 0), (3, 11, 33, 64, 192, 1), (2, 1, 70, 64, 64, 0), (2, 9, 1, 64, 64, 0),
    # the six-sweep kernel (algo 9, cout % 512 == 0): pixel-block quads that end inside the batch, 64 .. 192 input channels (2 .. 6 channel
    # groups per sweep), pooled odd maps, more quads than one XCD round
    (5, 14, 4, 512, 512, 0), (3, 29, 8, 192, 512, 1), (9, 9, 16, 64, 1024, 0), (40, 13, 7, 128, 512, 1), (2, 58, 16, 512, 512, 1),
    (3, 14, 4, 128, 1536, 0),   # 12 channel tiles (not a power of two)
]


@pytest.mark.parametrize("algo", [0, 1, 2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("n,H,W,cin,cout,pool", CONV_CASES)
def test_conv_layer_vs_torch(dev, n, H, W, cin, cout, pool, algo):
    from st_ito import _hip
    L = _hip.lib()
    g = torch.Generator().manual_seed(H * 1000 + W + cin)
    x = torch.randn((n, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)
    scale = 0.5 + torch.rand(cout, generator=g)
    shift = 0.2 * torch.randn(cout, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double()[None, :, None, None]
                     + shift.double()[None, :, None, None])
    if pool:
        ref = torch.nn.functional.avg_pool2d(ref, 2)
    def blocked(t):  # (n, C, H, W) -> channel-blocked (n, C/8, H, W, 8); C == 1 stays (n, H, W)
        n_, C_, H_, W_ = t.shape
        if C_ == 1:
            return t.reshape(n_, H_, W_).contiguous()
        return t.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()

    ref = blocked(ref)
    xd = blocked(x).to(dev)
    wd = w.contiguous().to(dev)
    if not L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo):
        assert algo != 0, "the direct kernel must cover every Cnn14-shaped layer"
        pytest.skip("shape not covered by the Winograd kernel (direct is used instead)")
    packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
    st = _hip.stream_ptr()
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(wd), cout, cin, algo, _hip.ptr(packed), st))
    out = torch.full(ref.shape, float("nan"), device=dev, dtype=torch.float32)
    sd, hd = scale.to(dev), shift.to(dev)
    wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, algo)
    assert (wsb > 0) == (algo in (3, 4, 5, 8, 9))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out),
                                          n, H, W, cin, cout, pool, algo, _hip.ptr(ws), wsb, st))
    got = out.cpu().double()
    assert not torch.isnan(got).any(), "unwritten outputs"
    if algo == 3:  # the hoisted input transform is the same arithmetic in the same order as algo 2: identical bits
        out2 = torch.full(ref.shape, float("nan"), device=dev, dtype=torch.float32)
        _hip.check(L.stito_conv3x3_bn_relu(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out2),
                                           n, H, W, cin, cout, pool, 2, st))
        assert torch.equal(out, out2)
        assert L.stito_conv3x3_bn_relu(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out2),
                                       n, H, W, cin, cout, pool, 3, st) == _hip.E_WORKSPACE  # no workspace, no launch
    err = (got - ref).abs().max().item()
    print(f"conv algo {algo} {n}x{H}x{W} {cin}->{cout} pool={pool}: max err {err:.3e} (ref max {ref.abs().max().item():.2f})")
    if algo in (4, 5, 8, 9):
        assert L.stito_conv3x3_bn_relu(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out),
                                       n, H, W, cin, cout, pool, algo, st) == _hip.E_WORKSPACE  # no workspace, no launch
    # F(4x4,3x3): random SIGNED inputs are the worst case for the cancellation in its output transform (3.3e-5 of the
    # maximum at cin = 2048); on real trunk activations it is as accurate as the direct kernel (tools/trunk_accuracy.py)
    tol = 5e-5 if algo >= 2 else 2e-5
    assert err < tol * max(1.0, ref.abs().max().item()), f"max err {err:.3e}"


@pytest.mark.parametrize("algo", [4, 5, 8, 9])
@pytest.mark.parametrize("n,H,W,cin,cout,pool", [(6, 14, 4, 512, 512, 0), (5, 29, 8, 256, 512, 1), (9, 58, 16, 64, 256, 0),
                                                 (7, 21, 32, 64, 64, 1), (12, 6, 16, 128, 64, 0), (13, 14, 4, 1024, 1024, 0)])
def test_conv_split_stream_scales(dev, n, H, W, cin, cout, pool, algo):
    """STITO_CONV_WINOGRAD_F4_SPLIT carries every operand as f16 hi + lo of a power-of-two multiple of itself; the
    multiple is chosen per stream from the stream's own largest activation.  Streams 1e4 and 1e-4 times the others, an
    all-zero stream and a stream with one huge outlier sit in ONE launch (several streams per workgroup tile on these
    maps): every stream must come out as accurate, relative to ITS OWN maximum, as the float32 kernels (no f16
    overflow, no loss on the quiet streams), and bitwise independent of what else is in the batch."""
    from st_ito import _hip
    L = _hip.lib()
    if (algo == 8 or cout < 256 or (algo == 9 and cout % 512)) and not L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo):
        pytest.skip("the register-resident kernel takes 64 input channels, the streaming ones outputs in multiples of 256 channels (six sweeps: 512)")
    assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo)
    g = torch.Generator().manual_seed(H * 7 + cin)
    x = torch.relu(torch.randn((n, cin, H, W), generator=g))
    x[1] *= 1e4
    x[2] *= 1e-4
    x[3] = 0.0
    x[4, cin // 2, H // 2, W // 2] = 3e3
    w = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)
    scale = 0.5 + torch.rand(cout, generator=g)
    shift = torch.zeros(cout)   # no shift: the outputs scale with the stream, so a per-stream relative bound is meaningful
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double()[None, :, None, None])
    if pool:
        ref = torch.nn.functional.avg_pool2d(ref, 2)
    def blocked(t):
        n_, C_, H_, W_ = t.shape
        return t.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()
    st = _hip.stream_ptr()
    wd, sd, hd = w.contiguous().to(dev), scale.to(dev), shift.to(dev)
    packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(wd), cout, cin, algo, _hip.ptr(packed), st))

    def run(xs):
        k = xs.shape[0]
        xd = blocked(xs).to(dev)
        out = torch.full((k, cout // 8) + tuple(ref.shape[2:]) + (8,), float("nan"), device=dev, dtype=torch.float32)
        wsb = L.stito_conv3x3_workspace_bytes(k, H, W, cin, cout, pool, algo)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out),
                                              k, H, W, cin, cout, pool, algo, _hip.ptr(ws), wsb, st))
        return out.cpu()
    got = run(x)
    refb = blocked(ref)
    for s_ in range(n):
        mx = refb[s_].abs().max().item()
        err = (got[s_].double() - refb[s_]).abs().max().item()
        print(f"stream {s_}: max {mx:.3e} err {err:.3e}")
        assert torch.isfinite(got[s_]).all()
        assert err <= 5e-5 * mx if mx > 0 else err == 0.0, (s_, err, mx)
    # batch independence: stream 2 (quiet) and stream 0 evaluated alone / in another order give the same bits
    alone = run(x[[2, 0]])
    assert torch.equal(alone[0], got[2]) and torch.equal(alone[1], got[0])


@pytest.mark.parametrize("n,H,W,cout,pool,wgs", [(6, 117, 64, 64, 1, 8), (3, 60, 100, 128, 0, 8), (5, 29, 128, 64, 0, 16), (2, 469, 128, 64, 1, 24)])
def test_conv_f2reg_persistent_loop(dev, n, H, W, cout, pool, wgs, monkeypatch):
    """STITO_CONV_WINOGRAD_F2_REG runs persistent workgroups (one per CU) that walk over pixel groups through a 7-entry ring of
    patch k-steps (4 per group): with the grid cut down to a few workgroups (STITO_W23_WG, a tuning aid) every workgroup
    walks tens to hundreds of groups, so every alignment of a group in the ring, every edge-mask change between consecutive
    groups (left / right / top / bottom of the map, stream changes) and the hand-off of a group's outputs to the next trip are
    exercised; results must equal the default grid's bit for bit (a pixel group's arithmetic does not depend on who computes
    it) and agree with float64 torch."""
    from st_ito import _hip
    L = _hip.lib()
    cin, algo = 64, 8
    g = torch.Generator().manual_seed(H + W + cout)
    x = torch.relu(torch.randn((n, cin, H, W), generator=g))
    w = torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)
    scale = 0.5 + torch.rand(cout, generator=g)
    shift = 0.2 * torch.randn(cout, generator=g)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), padding=1) * scale.double()[None, :, None, None]
                     + shift.double()[None, :, None, None])
    if pool:
        ref = torch.nn.functional.avg_pool2d(ref, 2)
    def blocked(t):
        n_, C_, H_, W_ = t.shape
        return t.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()
    ref = blocked(ref)
    xd, wd, sd, hd = blocked(x).to(dev), w.contiguous().to(dev), scale.to(dev), shift.to(dev)
    st = _hip.stream_ptr()
    assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo)
    packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(wd), cout, cin, algo, _hip.ptr(packed), st))
    wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, algo)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    outs = []
    for env in (str(wgs), None):
        if env is None:
            monkeypatch.delenv("STITO_W23_WG", raising=False)
        else:
            monkeypatch.setenv("STITO_W23_WG", env)
        out = torch.full(ref.shape, float("nan"), device=dev, dtype=torch.float32)
        for _ in range(3):  # repeated launches into the same buffers
            _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(xd), _hip.ptr(packed), _hip.ptr(sd), _hip.ptr(hd), _hip.ptr(out),
                                                  n, H, W, cin, cout, pool, algo, _hip.ptr(ws), wsb, st))
        outs.append(out.cpu())
    err = (outs[0].double() - ref).abs().max().item()
    print(f"f2reg {n}x{H}x{W} 64->{cout} pool={pool} wgs={wgs}: max err {err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert not torch.isnan(outs[0]).any() and not torch.isnan(outs[1]).any(), "unwritten outputs"
    assert err < 5e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(outs[0], outs[1])


# (H, W, cin, cout, pool, algorithms): the deep layers where the streaming kernels keep three slabs of LDS-DMA copies in flight
# (the copies complete out of issue order: csrc/conv_wino43.hip header), and the persistent register-resident kernel
RACE_CASES = [(14, 4, 2048, 2048, 0, (2, 3, 5, 9)), (29, 8, 512, 1024, 1, (2, 3, 5, 9)), (58, 16, 512, 512, 1, (2, 3, 4, 5, 9)),
              (117, 32, 256, 256, 1, (2, 4)), (117, 32, 128, 256, 0, (2, 4)), (234, 64, 64, 128, 0, (2, 8)), (469, 128, 64, 64, 1, (8,))]


@pytest.mark.parametrize("H,W,cin,cout,pool,algos", RACE_CASES)
def test_conv_repeat_launch_race_guard(dev, H, W, cin, cout, pool, algos):
    """Every shipped conv algorithm on the bench's layer shapes at the bench's batch (512 streams), 10 launches each into
    NaN-prefilled outputs: all ten results bitwise equal (a workgroup that consumed a slab before its LDS-DMA copy had landed --
    the failure tools/conv_stress.py was written to hunt -- shows as a launch that differs), no NaN left (unwritten outputs), and
    within 5e-5 of the output maximum of the direct float32 kernel on the same data."""
    from st_ito import _hip
    L = _hip.lib()
    n, reps = 512, 10
    st = _hip.stream_ptr()
    g = torch.Generator(device="cpu").manual_seed(H + cin)
    x = torch.relu(torch.randn((n, cin // 8, H, W, 8), generator=g)).to(dev)
    w = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).to(dev)
    sc, sh = (0.5 + torch.rand(cout, generator=g)).to(dev), (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)

    def run(algo, out):
        packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, algo, _hip.ptr(packed), st))
        wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, algo)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        outs = []
        for _ in range(out):
            o = torch.full((n, cout // 8, Ho, Wo, 8), float("nan"), device=dev)
            _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(o), n, H, W, cin,
                                                  cout, pool, algo, _hip.ptr(ws), wsb, st))
            outs.append(o)
        return outs
    ref = run(0, 1)[0]
    assert not torch.isnan(ref).any()
    tol = 5e-5 * max(1.0, ref.abs().max().item())
    for algo in algos:
        assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo), (algo, H, W, cin, cout)
        outs = run(algo, reps)
        assert not torch.isnan(outs[0]).any(), f"algo {algo}: unwritten outputs"
        err = (outs[0] - ref).abs().max().item()
        print(f"race guard {H}x{W} {cin}->{cout} pool={pool} algo {algo}: {reps} launches, max |diff| to the direct kernel {err:.3e} (bound {tol:.3e})")
        assert err < tol, (algo, err)
        for r in range(1, reps):
            assert torch.equal(outs[r], outs[0]), f"algo {algo}: launch {r} differs from launch 0"
        del outs


@pytest.mark.parametrize("n,H,W,cout,pool", [(2, 33, 128, 64, 1), (3, 9, 64, 64, 1), (5, 6, 32, 128, 1), (2, 21, 40, 128, 0),
                                              (1, 469, 128, 64, 1), (4, 13, 128, 64, 1), (3, 7, 130, 64, 0), (6, 4, 16, 64, 1),
                                              (2, 2, 2, 64, 1), (1, 1, 1, 64, 0), (40, 10, 34, 64, 1)])
def test_conv_block1_f2reg_vs_torch(dev, n, H, W, cout, pool):
    """stito_conv_block1_f2reg (ABI v8; the model's default for conv_block1): relu(bn1(conv3x3(x))) -> relu(bn2(conv3x3(.)))
    (+ 2x2 average pool) of a 1-channel map in ONE launch of the register-resident F(2x2,3x3) kernel -- the first conv is
    computed on the f16 matrix pipe (split operands) into the second conv's patch ring (panns.py:65-80, 250) -- against
    float64 torch, PER STREAM (5e-5 of the stream's own output maximum), with streams 1e3 x / 1e-3 x the others, an all-zero
    stream and a stream with one outlier in one launch; bitwise equal for every persistent-grid size (a workgroup's walk over
    its pixel groups, the look-ahead behind its last group), and the reported per-stream maxima equal those of the output."""
    from st_ito import _hip
    L = _hip.lib()
    c1 = 64
    assert L.stito_conv_block1_f2reg_supported(n, H, W, c1, cout, pool)
    g = torch.Generator().manual_seed(H * 100 + W + cout)
    x = torch.randn((n, 1, H, W), generator=g)
    if n >= 2: x[1] *= 1e3
    if n >= 3: x[2] *= 1e-3
    if n >= 4: x[3] = 0.0
    if n >= 5: x[4, 0, H // 2, W // 3] = 250.0
    w1 = torch.randn((c1, 1, 3, 3), generator=g) / 3.0
    w2 = torch.randn((cout, c1, 3, 3), generator=g) / np.sqrt(9 * c1)
    s1, h1 = 0.5 + torch.rand(c1, generator=g), 0.3 * torch.randn(c1, generator=g)
    s2, h2 = 0.5 + torch.rand(cout, generator=g), 0.2 * torch.randn(cout, generator=g)
    F = torch.nn.functional
    y = torch.relu(F.conv2d(x.double(), w1.double(), padding=1) * s1.double()[None, :, None, None] + h1.double()[None, :, None, None])
    y = torch.relu(F.conv2d(y, w2.double(), padding=1) * s2.double()[None, :, None, None] + h2.double()[None, :, None, None])
    if pool:
        y = F.avg_pool2d(y, 2)
    n_, C_, H_, W_ = y.shape
    ref = y.reshape(n_, C_ // 8, 8, H_, W_).permute(0, 1, 3, 4, 2).contiguous()
    st = _hip.stream_ptr()
    xd = x.reshape(n, H, W).contiguous().to(dev)
    w1d, s1d, h1d, w2d, s2d, h2d = (t.contiguous().to(dev) for t in (w1, s1, h1, w2, s2, h2))
    fw = torch.empty(L.stito_cnn14_packed_conv1_f2reg_floats(), device=dev)
    _hip.check(L.stito_cnn14_pack_conv1_f2reg(_hip.ptr(w1d), _hip.ptr(s1d), _hip.ptr(h1d), c1, _hip.ptr(fw), st))
    upk = torch.empty(L.stito_cnn14_packed_conv_floats(cout, c1, _hip.CONV_WINOGRAD_F2_REG), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w2d), cout, c1, _hip.CONV_WINOGRAD_F2_REG, _hip.ptr(upk), st))
    wsb = L.stito_conv_block1_f2reg_workspace_bytes(n, H, W, c1, cout, pool)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def run():
        out = torch.full(ref.shape, float("nan"), device=dev, dtype=torch.float32)
        amax = torch.zeros(n, dtype=torch.int32, device=dev)
        _hip.check(L.stito_conv_block1_f2reg(_hip.ptr(xd), _hip.ptr(fw), _hip.ptr(upk), _hip.ptr(s2d), _hip.ptr(h2d), _hip.ptr(out),
                                             n, H, W, c1, cout, pool, _hip.ptr(ws), wsb, st, _hip.ptr(amax)))
        return out, amax

    out, amax = run()
    got = out.cpu().double()
    assert not torch.isnan(got).any(), "unwritten outputs"
    worst = 0.0
    for i in range(n):
        m = ref[i].abs().max().item()
        e = (got[i] - ref[i]).abs().max().item()
        if H_ * W_ > 0:
            worst = max(worst, e / max(m, 1e-30))
            assert e <= 5e-5 * m + 1e-30, f"stream {i}: max err {e:.3e} of {m:.3e}"
    print(f"fused block (f2reg) {n}x{H}x{W} 1->{c1}->{cout} pool={pool}: worst per-stream err {worst:.2e} of the stream's maximum")
    if H_ * W_ > 0:
        assert torch.equal(amax.cpu().view(torch.float32), out.reshape(n, -1).max(dim=1).values.cpu()), "reported per-stream maxima"
    # against the two launches (first conv kernel, then the same F(2x2,3x3) kernel on its stored output): the same function
    pk1 = torch.empty(L.stito_cnn14_packed_conv_floats(c1, 1, 0), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w1d), c1, 1, 0, _hip.ptr(pk1), st))
    mid = torch.empty((n, c1 // 8, H, W, 8), device=dev)
    _hip.check(L.stito_conv3x3_bn_relu(_hip.ptr(xd), _hip.ptr(pk1), _hip.ptr(s1d), _hip.ptr(h1d), _hip.ptr(mid), n, H, W, 1, c1, 0, 0, st))
    out2 = torch.empty_like(out)
    wsb2 = L.stito_conv3x3_workspace_bytes(n, H, W, c1, cout, pool, _hip.CONV_WINOGRAD_F2_REG)
    ws2 = torch.empty(max(wsb2, 16), dtype=torch.uint8, device=dev)
    _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(mid), _hip.ptr(upk), _hip.ptr(s2d), _hip.ptr(h2d), _hip.ptr(out2), n, H, W, c1, cout, pool,
                                          _hip.CONV_WINOGRAD_F2_REG, _hip.ptr(ws2), wsb2, st))
    for i in range(n):
        m = ref[i].abs().max().item()
        assert (out[i] - out2[i]).abs().max().item() <= 2e-5 * m + 1e-30, f"stream {i}: fused vs two launches"
    # persistent loop: any number of workgroups per channel block gives the same bits
    for wg in ("8", "16", "64"):
        os.environ["STITO_W23_WG"] = wg
        try:
            o2, a2 = run()
        finally:
            del os.environ["STITO_W23_WG"]
        assert torch.equal(o2, out) and torch.equal(a2, amax), f"STITO_W23_WG={wg}"


@pytest.mark.parametrize("norm", ["minmax", "batchnorm", "none"])
def test_model_vs_golden_reference(dev, golden_dir, norm):
    """Full Cnn14 forward + get_param_embeds against the reference's conv stack output."""
    from st_ito.utils import get_param_embeds
    g = np.load(os.path.join(golden_dir, f"cnn14_trunk_{norm}.npz"))
    om, pm = _models(dev, norm)
    x = torch.from_numpy(g["x"])
    mid, side = pm(x.to(dev))
    for got, key in ((mid, "mid"), (side, "side")):
        ref = g[key]
        rel = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert rel < 1e-4, f"{key}: {rel:.3e}"
    midm, sidem = pm(torch.from_numpy(g["x_mono"]).to(dev))
    assert np.abs(midm.cpu().numpy() - g["mid_mono"]).max() / np.abs(g["mid_mono"]).max() < 1e-4
    assert torch.equal(midm, sidem)
    e = get_param_embeds(x.clone(), pm, SR)
    assert e["mid"].device == x.device and e["mid"].dtype == x.dtype
    for key in ("mid", "side"):
        ref = g[f"embed_{key}"]
        assert np.abs(e[key].numpy() - ref).max() / np.abs(ref).max() < 1e-4


def test_model_vs_golden_reference_bench_shaped_map(dev, golden_dir):
    """The reference's OWN conv stack on a 257 x 128 log-mel map (n = 262 144, the CLI's default crop; final map 8 x 4):
    tests/golden/cnn14_trunk_minmax_262144.npz (make_golden.py g6b) -- the small fixtures above only reach a 1 x 4 final
    map.  Raw fc outputs and get_param_embeds within 1e-4 of the reference's values."""
    from st_ito.utils import get_param_embeds
    g = np.load(os.path.join(golden_dir, "cnn14_trunk_minmax_262144.npz"))
    _, pm = _models(dev, "minmax", int(g["seed"]))
    x = O.synth_audio(int(g["audio_seed"]), 2, int(g["n"]))[None]
    mid, side = pm(x.to(dev))
    for got, key in ((mid, "mid"), (side, "side")):
        rel = np.abs(got.cpu().numpy() - g[key]).max() / np.abs(g[key]).max()
        print(f"{key}: {rel:.2e} of the reference's maximum")
        assert rel < 1e-4, f"{key}: {rel:.3e}"
    e = get_param_embeds(x.clone(), pm, SR)
    for key in ("mid", "side"):
        ref = g[f"embed_{key}"]
        assert np.abs(e[key].numpy() - ref).max() / np.abs(ref).max() < 1e-4


def test_trunk_hoisted_input_transform_is_bitwise_the_in_kernel_one(dev):
    """The float32 trunk (conv_split = False): F(4x4,3x3) with the input transform hoisted into its own pass from 512
    output channels up (conv_block4-6, STITO_CONV_WINOGRAD_F4_PRE) against the same trunk with every layer transforming
    in-kernel (conv_pre_min_cout = 0): same arithmetic in the same order, so the embeddings are identical bit for bit -- on
    a bench-shaped input (10 s: the 469 x 128 map, all tile widths 8 / 4 / 2 / 1) and on a short one (ragged tiles).
    The model's default (conv_split: the layers from 256 output channels up on the f16 matrix pipe with split operands,
    STITO_CONV_WINOGRAD_F4_SPLIT) is not bitwise the float32 trunk; it must sit inside float32 rounding of it."""
    from st_ito import _hip
    from st_ito.models.panns import Cnn14
    om = O.fill_deterministic(O.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "batchnorm"), 0).eval()
    outs = {}
    for pre in (512, 0, "split"):
        pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "batchnorm")
        pm.load_state_dict(om.state_dict())
        pm.eval().to(dev)
        if pre == "split":
            assert pm.conv_split and pm.conv_split_min_cout == 256   # the defaults
        else:
            pm.conv_split = False
            pm.conv_pre_min_cout = pre
        W, _, _ = pm._ensure()
        algos = [int(W.conv_wino_algo[i]) for i in range(12)]
        if pre == "split":
            assert algos[1:3] == [_hip.CONV_WINOGRAD_F2_REG] * 2 and algos[3] == _hip.CONV_WINOGRAD_F4   # the 64-input-channel layers: register-resident F(2x2,3x3)
            assert algos[4:7] == [_hip.CONV_WINOGRAD_F4_SPLIT] * 3 and algos[7:] == [_hip.CONV_WINOGRAD_F4_SPLIT3] * 5   # from 512 input channels: 128 x 128 tiles in six sweeps
        else:
            assert algos[1:6] == [_hip.CONV_WINOGRAD_F4] * 5
            assert algos[6:] == [_hip.CONV_WINOGRAD_F4_PRE if pre else _hip.CONV_WINOGRAD_F4] * 6
        for n in (480000, 40001):
            x = torch.stack([O.synth_audio(70 + i, 2, n) for i in range(3)])
            outs[(pre, n)] = [t.clone() for t in pm(x.to(dev))]
    for n in (480000, 40001):
        for a, b, c in zip(outs[(512, n)], outs[(0, n)], outs[("split", n)]):
            assert torch.equal(a, b)
            rel = ((c - a).abs().max() / a.abs().max()).item()
            print(f"split-precision trunk vs float32 trunk, n = {n}: {rel:.2e} of the embedding maximum")
            assert rel < 5e-6, rel


@pytest.mark.parametrize("n,H,W,cin,cout,pool", [(5, 14, 4, 512, 512, 0), (3, 29, 8, 192, 512, 1), (9, 9, 16, 64, 1024, 0), (40, 13, 7, 128, 512, 1),
                                                 (2, 58, 16, 512, 512, 1), (64, 14, 4, 1024, 1024, 0), (7, 14, 4, 2048, 2048, 0),
                                                 # cout 1536 = 12 channel tiles: not a power of two (the round-5 grid came out empty for it),
                                                 # with few and with many pixel-block quads (xm = 8 / 4 / 2 splits of the XCDs)
                                                 (3, 14, 4, 128, 1536, 0), (80, 14, 4, 64, 1536, 1), (300, 9, 4, 64, 1536, 0)])
def test_conv_six_sweeps_is_bitwise_the_two_sweep_kernel(dev, n, H, W, cin, cout, pool):
    """k_conv_wino43s3 (128 x 128 tiles, six sweeps) and k_conv_wino43s2 (64 x 64, two sweeps) do the same arithmetic in the same
    order -- per position the same sequence of products over the input channels (input lo x weight hi, hi x lo, hi x hi per 16
    channels; an MFMA's sum over k does not depend on which operand carries which), then Y = A^T M A with the same association --
    so which of the two a batch size selects (stito_cnn14_forward takes the two-sweep packing for small batches) cannot be seen
    in the results: a candidate's fitness stays bitwise independent of the batch it is evaluated in."""
    from st_ito import _hip
    L = _hip.lib()
    st = _hip.stream_ptr()
    g = torch.Generator().manual_seed(H * 10 + cin)
    x = torch.relu(torch.randn((n, cin // 8, H, W, 8), generator=g)).to(dev)
    x[1 % n] *= 300.0
    w = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).to(dev)
    sc = (0.5 + torch.rand(cout, generator=g)).to(dev); sh = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    outs = {}
    for m in (_hip.CONV_WINOGRAD_F4_SPLIT2, _hip.CONV_WINOGRAD_F4_SPLIT3):
        assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, m)
        packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, m), device=dev)
        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, m, _hip.ptr(packed), st))
        out = torch.full((n, cout // 8, Ho, Wo, 8), float("nan"), device=dev)
        wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, m)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), n, H, W, cin, cout, pool, m,
                                              _hip.ptr(ws), wsb, st))
        outs[m] = out
    assert not torch.isnan(outs[_hip.CONV_WINOGRAD_F4_SPLIT3]).any()
    assert torch.equal(outs[_hip.CONV_WINOGRAD_F4_SPLIT2], outs[_hip.CONV_WINOGRAD_F4_SPLIT3])


@pytest.mark.parametrize("n,H,W,cin,cout,pool", [(64, 8, 4, 2048, 2048, 0), (64, 16, 8, 1024, 1024, 1), (5, 14, 4, 512, 512, 0), (3, 29, 8, 192, 512, 1),
                                                 (40, 13, 7, 128, 256, 1), (9, 9, 16, 64, 1024, 0), (2, 58, 16, 512, 768, 1)])
def test_conv_sweep_split_and_xcd_order_are_bitwise_neutral(dev, n, H, W, cin, cout, pool, monkeypatch):
    """Round 6, small batches: the streaming kernels with the sweeps of an item as SEPARATE workgroups --
    k_conv_wino43s2: two (the second waits for the first one's flag before it reads the partial outputs), k_conv_wino43s3: six
    (the row-5 workgroup waits for a count of five) -- and k_conv_wino43s2 with the XCDs split in two dimensions must return the
    bits of the one-workgroup launches: same loads, same products, same additions in the same order.  Ten launches per setting (a
    lost flag / an early read of the partials would show as a difference, or as a hang -> run under `timeout`); the first two shapes
    are conv_block6 / conv_block5.conv2 of a population of 32 on 262 144 samples.  The two kernels also agree with each other."""
    from st_ito import _hip
    L = _hip.lib()
    st = _hip.stream_ptr()
    g = torch.Generator().manual_seed(H * 10 + cin)
    x = torch.relu(torch.randn((n, cin // 8, H, W, 8), generator=g)).to(dev)
    x[1 % n] *= 300.0
    w = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).to(dev)
    sc = (0.5 + torch.rand(cout, generator=g)).to(dev); sh = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = None
    settings = [(_hip.CONV_WINOGRAD_F4_SPLIT2, {"STITO_W43S2_SWSPLIT": sp, "STITO_W43S2_XM": xm})
                for sp, xm in (("0", "8"), ("1", "8"), ("1", None), ("0", None), ("1", "2"), ("1", "1"), ("0", "4"))]
    if cout % 512 == 0:
        settings += [(_hip.CONV_WINOGRAD_F4_SPLIT3, {"STITO_W43S3_SWSPLIT": sp}) for sp in ("0", "1", None)]
    for m, env in settings:
        for k in ("STITO_W43S2_SWSPLIT", "STITO_W43S2_XM", "STITO_W43S3_SWSPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            if v is not None:
                monkeypatch.setenv(k, v)
        assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, m)
        packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, m), device=dev)
        _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, m, _hip.ptr(packed), st))
        wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, m)   # the grid (and with it the scratch area) follows the XCD order
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        for rep in range(10):
            out = torch.full((n, cout // 8, Ho, Wo, 8), float("nan"), device=dev)
            _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), n, H, W, cin, cout, pool, m,
                                                  _hip.ptr(ws), wsb, st))
            if ref is None:
                ref = out
                assert not torch.isnan(ref).any()
            assert torch.equal(out, ref), (m, env, rep)


@pytest.mark.parametrize("algo,n,H,W,cin,cout,pool", [(4, 64, 117, 32, 128, 256, 0), (4, 40, 117, 32, 256, 256, 1), (2, 24, 234, 64, 128, 128, 1)])
def test_conv_layer_as_several_grids_on_several_queues_is_bitwise_one_grid(dev, algo, n, H, W, cin, cout, pool, monkeypatch):
    """Round 6: k_conv_wino43s launches a layer as TWO grids on two streams (two hardware queues overlap the hand-over between the
    one-per-CU workgroups: bench step - 0.4 ms; the f32 kernel can do the same, STITO_W43_QUEUES_F32).  An item does not know which
    grid it rode in: 1, 2, 3 and 4 queues must return the same bits, launch after launch (the side streams fork from and join the
    caller's stream by events: a missing join would show as a stale or torn output)."""
    from st_ito import _hip
    L = _hip.lib()
    st = _hip.stream_ptr()
    g = torch.Generator().manual_seed(H + cin)
    x = torch.relu(torch.randn((n, cin // 8, H, W, 8), generator=g)).to(dev)
    w = (torch.randn((cout, cin, 3, 3), generator=g) / np.sqrt(9 * cin)).to(dev)
    sc = (0.5 + torch.rand(cout, generator=g)).to(dev); sh = (0.1 * torch.randn(cout, generator=g)).to(dev)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    assert L.stito_conv3x3_supported(n, H, W, cin, cout, pool, algo)
    packed = torch.empty(L.stito_cnn14_packed_conv_floats(cout, cin, algo), device=dev)
    _hip.check(L.stito_cnn14_pack_conv(_hip.ptr(w), cout, cin, algo, _hip.ptr(packed), st))
    wsb = L.stito_conv3x3_workspace_bytes(n, H, W, cin, cout, pool, algo)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ref = None
    for q in ("1", "2", "3", "4", "2"):
        monkeypatch.setenv("STITO_W43_QUEUES", q)
        monkeypatch.setenv("STITO_W43_QUEUES_F32", q)
        for rep in range(5):
            out = torch.full((n, cout // 8, Ho, Wo, 8), float("nan"), device=dev)
            _hip.check(L.stito_conv3x3_bn_relu_ws(_hip.ptr(x), _hip.ptr(packed), _hip.ptr(sc), _hip.ptr(sh), _hip.ptr(out), n, H, W, cin, cout, pool, algo,
                                                  _hip.ptr(ws), wsb, st))
            x2 = out.clone()   # (a consumer on the caller's stream right behind the launch: it must see the joined result)
            if ref is None:
                ref = x2
                assert not torch.isnan(ref).any()
            assert torch.equal(x2, ref), (q, rep)


def test_trunk_small_batches_take_the_two_sweep_packing(dev):
    """The six-sweep kernel's workgroups are four times the two-sweep kernel's: a batch that gives it fewer than 3 / 4 workgroup per
    CU (anything below ~380 streams for conv_block6) runs the two-sweep kernel from the alternative packing the model carries
    (stito_cnn14_weights.conv_alt_dev, ABI v9) -- bitwise what a model without the six-sweep packing computes."""
    from st_ito import _hip
    from st_ito.models.panns import Cnn14
    om = O.fill_deterministic(O.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "batchnorm"), 0).eval()
    outs = {}
    for kind in ("default", "two-sweep only"):
        pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "batchnorm")
        pm.load_state_dict(om.state_dict())
        pm.eval().to(dev)
        if kind != "default":
            pm.conv_split3_min_cin = 0
        W, _, _ = pm._ensure()
        for i in range(7, 12):
            if kind == "default":
                assert int(W.conv_wino_algo[i]) == _hip.CONV_WINOGRAD_F4_SPLIT3 and W.conv_alt_dev[i] and int(W.conv_alt_algo[i]) == _hip.CONV_WINOGRAD_F4_SPLIT2
            else:
                assert int(W.conv_wino_algo[i]) == _hip.CONV_WINOGRAD_F4_SPLIT2 and not W.conv_alt_dev[i]
        x = torch.stack([O.synth_audio(70 + i, 2, 240000) for i in range(4)])
        outs[kind] = [t.clone() for t in pm(x.to(dev))]
        base = [O.synth_audio(90 + k, 2, 480000) for k in range(5)]
        xl = torch.stack([base[i % 5] * (1.0 + 0.01 * i) for i in range(200)])   # 400 streams: the six-sweep kernel runs
        outs[kind + " large"] = [t.clone() for t in pm(xl.to(dev))]
    for a, b in zip(outs["default"], outs["two-sweep only"]):
        assert torch.equal(a, b)
    for a, b in zip(outs["default large"], outs["two-sweep only large"]):   # ... and computes the same bits (test_conv_six_sweeps_is_bitwise_the_two_sweep_kernel)
        assert torch.equal(a, b)


@pytest.mark.parametrize("first,last,chunk", [(2, 6, 4), (2, 6, 6), (0, 6, 4), (2, 5, 4), (3, 7, 4), (4, 4, 4), (0, 11, 3), (2, 11, 2), (1, 2, 4), (3, 6, 4)])
def test_trunk_depth_first_chunks_are_bitwise_the_layer_by_layer_schedule(dev, first, last, chunk):
    """stito_cnn14_weights.chunk_* (ABI v10): a run of convs scheduled depth-first over chunks of streams (round 6: the hand-offs
    of a chunk stay in the Infinity Cache) returns the embeddings of the layer-by-layer order bit for bit -- chunk sizes that do
    and do not divide the batch (14 streams), runs that start / end on either conv of a block, the run with conv_block1's single
    launch inside, one layer alone, a run that starts on conv_block1's second conv (it takes the whole one-launch block), and a run the forward
    refuses to schedule ((3, 6): a mid-block start ending on a first conv -- input and output would share a buffer -- runs layer by
    layer) -- and every layer is still timed (launch count = layers x chunks)."""
    from st_ito import _hip
    from st_ito.models.panns import Cnn14
    L = _hip.lib()
    om = O.fill_deterministic(O.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "minmax"), 0).eval()
    base = [O.synth_audio(50 + k, 2, 120000) for k in range(3)]
    x = torch.stack([base[i % 3] * (1.0 if i != 2 else 1e-3) * (1.0 + 0.05 * i) for i in range(7)]).to(dev)   # 14 streams
    outs, counts = {}, {}
    for kind in ("layers", "chunks"):
        pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "minmax")
        pm.load_state_dict(om.state_dict())
        pm.eval().to(dev)
        pm.trunk_chunk, pm.trunk_chunk_convs = (chunk, (first, last)) if kind == "chunks" else (0, (2, 6))
        _hip.check(L.stito_conv_timing_enable(1))
        try:
            outs[kind] = [t.clone() for t in pm(x)]
            ms, tag, cnt = (ctypes.c_double * 512)(), (ctypes.c_int * 512)(), ctypes.c_int()
            _hip.check(L.stito_conv_timing_read_tagged(ms, tag, 512, ctypes.byref(cnt)))
        finally:
            _hip.check(L.stito_conv_timing_enable(0))
        counts[kind] = [list(tag[: cnt.value]).count(i) for i in range(12)]
    assert counts["layers"] == [0] + [1] * 11, counts
    n_chunks = -(-14 // chunk)
    if first == 1:
        first = 0   # conv_block1 is one launch
    unscheduled = (first % 2 == 1 and last % 2 == 0 and first != last)
    lo = 1 if first == 0 else first
    want = [0] + [n_chunks if (lo <= i <= last and not unscheduled) else 1 for i in range(1, 12)]
    assert counts["chunks"] == want, (counts, want)
    for a, b in zip(outs["layers"], outs["chunks"]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("norm", ["batchnorm", "minmax", "none"])
def test_trunk_block1_one_launch_is_the_default_and_matches_two_launches(dev, norm):
    """The default trunk runs conv_block1 as ONE launch (conv1_f2reg_w_dev set -> stito_cnn14_forward calls
    stito_conv_block1_f2reg; 11 timed conv launches per pass either way: the first conv on its own is VALU work and not
    timed); STITO_CONV_FUSE1=0 / conv_fuse1 = False restores the two launches.
    Same function: embeddings inside float32 rounding of each other on a bench-shaped input and a short one, for every
    input_norm mode (raw dB values up to ~100 included), with a stream 1e-3 x the others in the batch."""
    from st_ito import _hip
    from st_ito.models.panns import Cnn14
    L = _hip.lib()
    om = O.fill_deterministic(O.Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, norm), 0).eval()
    outs, launches = {}, {}
    for kind in ("one", "two"):
        pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, norm)
        pm.load_state_dict(om.state_dict())
        pm.eval().to(dev)
        if kind == "two":
            pm.conv_fuse1 = False
        W, _, _ = pm._ensure()
        assert bool(W.conv1_f2reg_w_dev) == (kind == "one")
        assert int(W.conv_wino_algo[1]) == _hip.CONV_WINOGRAD_F2_REG
        for n in (480000, 40001):
            x = torch.stack([O.synth_audio(70 + i, 2, n) * (1.0 if i != 1 else 1e-3) for i in range(3)])
            _hip.check(L.stito_conv_timing_enable(1))
            try:
                outs[(kind, n)] = [t.clone() for t in pm(x.to(dev))]
                tot, cnt = ctypes.c_double(), ctypes.c_int()
                _hip.check(L.stito_conv_timing_read(ctypes.byref(tot), ctypes.byref(cnt)))
            finally:
                _hip.check(L.stito_conv_timing_enable(0))
            launches[(kind, n)] = cnt.value
    for n in (480000, 40001):
        assert launches[("one", n)] == 11 and launches[("two", n)] == 11, launches
        for a, c in zip(outs[("two", n)], outs[("one", n)]):
            rel = ((c - a).abs().max() / a.abs().max()).item()
            print(f"conv_block1 in one launch vs two, input_norm {norm}, n = {n}: {rel:.2e} of the embedding maximum")
            assert rel < 5e-6, rel


@pytest.mark.parametrize("tag", ["stereo", "mono"])
def test_evaluate_losses_golden(dev, golden_dir, tag):
    """run_es.evaluate end to end (pad to 262144, render, embed, cosine) against the losses the
    reference's own run_es produced for the same population (tests/golden/evaluate_*.npz)."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds
    from st_ito.style_transfer import process_audio
    g = np.load(os.path.join(golden_dir, f"evaluate_{tag}.npz"))
    _, pm = _models(dev, "minmax", int(g["seed"]))
    pp = E.make_plugins("eq")
    x = torch.from_numpy(g["x"].copy()); tgt = torch.from_numpy(g["target"].copy())
    x /= x.abs().max().clamp(min=1e-8); tgt /= tgt.abs().max().clamp(min=1e-8)
    te = get_param_embeds(tgt, pm, SR)
    ev = PopulationEvaluator(x, SR, pp, pm, te)
    loss, embeds, audio = ev.evaluate(list(g["W"]), want_audio=True)
    fv = loss.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(fv, g["fvals"], rtol=1e-4, atol=2e-6)
    assert audio.shape[-1] == 262144 and audio.shape[0] == len(g["W"])
    assert int(np.argmin(fv)) == int(np.argmin(g["fvals"]))
    out = process_audio(x.squeeze(0).numpy(), g["wopt"], SR, pp)
    np.testing.assert_allclose(out, g["output_audio"], rtol=0, atol=3e-6)


def test_full_chain_losses_vs_oracle(dev):
    """BASELINE chain (EQ/comp/reverb/EQ/gain), stereo: per-candidate losses HIP vs oracle."""
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds
    chain = ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]
    op, pp = _plugins_pair(chain)
    om, pm = _models(dev, "minmax")
    rng = np.random.default_rng(8)
    n, P = 300000, 4
    x = O.synth_audio(41, 2, n)[None]
    D = sum(p["num_params"] for p in op.values())
    assert D == 45
    wt = np.random.default_rng(7).random(D)
    tgt = torch.from_numpy(O.process_audio(x[0].numpy().copy(), wt, SR, op))[None]
    W = rng.random((P, D))
    te_o = O.get_param_embeds(tgt.clone(), om, SR)
    f_ref, e_ref, _ = O.evaluate(list(W), x, SR, op, te_o, om)
    te = get_param_embeds(tgt.clone(), pm, SR)
    for k in ("mid", "side"):
        assert (te[k] - te_o[k]).abs().max() / te_o[k].abs().max() < 1e-4
    ev = PopulationEvaluator(x, SR, pp, pm, te)
    loss, embeds, _ = ev.evaluate(list(W))
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(f_ref), rtol=1e-4, atol=5e-6)
    for k in ("mid", "side"):
        rel = (embeds[k].cpu() - e_ref[k]).abs().max() / e_ref[k].abs().max()
        assert rel < 1e-4, f"{k} embeddings: {rel:.3e}"


def test_process_audio_normalize_stages_vs_oracle(dev):
    """process_audio(normalize_stages=True), style_transfer.py:106-107: joint peak normalisation after
    every plugin.  Mono input through the run_optim 'basic' chain: the first three stages are mono
    inside a stereo result buffer (strided peak scan), then the delay up-mixes."""
    from st_ito.style_transfer import process_audio
    op, pp = _plugins_pair(["ParametricEQ", "Compressor", "Distortion", "Delay", "Reverb"])
    rng = np.random.default_rng(12)
    for chs, n in ((1, 30000), (2, 24001)):
        x = O.synth_audio(30 + chs, chs, n).numpy()
        w = rng.random(31)
        ref = O.process_audio(x.copy(), w, SR, op, normalize_stages=True)
        got = process_audio(x.copy(), w, SR, pp, normalize_stages=True)
        plain = process_audio(x.copy(), w, SR, pp)
        assert got.shape == ref.shape == (2, n)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
        assert np.abs(got - plain).max() > 1e-3  # the compressor / distortion see a different level: it is not a no-op


def test_features_vs_reference_golden_and_oracle(dev, golden_dir):
    """st_ito/features.py on the GPU against the vectors produced by the reference's own code (G8:
    bark spectrum in three modes and two FFT sizes, RMS, crest factor) and against the oracle
    (spectral centroid: torchaudio restatement, unpinned; LUFS: host)."""
    from st_ito import features as PF
    g = np.load(os.path.join(golden_dir, "features.npz"))
    x = torch.stack([O.synth_audio(int(sd), 2, int(g["n"])) * float(sc) for sd, sc in zip(g["seeds"], g["scales"])])
    for fft in (32768, 4096):
        for mode in ("mono", "stereo", "mid-side"):
            got = PF.compute_barkspectrum(x, fft_size=fft, sample_rate=SR, mode=mode)
            assert got.device == x.device and got.shape == (3, 24 if mode == "mono" else 48)
            np.testing.assert_allclose(got.numpy(), g[f"bark_{fft}_{mode.replace('-', '')}"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(PF.compute_rms_energy(x).numpy(), g["rms"], rtol=2e-6)
    np.testing.assert_allclose(PF.compute_crest_factor(x).numpy(), g["crest"], rtol=0, atol=2e-5)
    xm = x[:, :1]
    np.testing.assert_allclose(PF.compute_barkspectrum(xm, sample_rate=SR, mode="mono").numpy(),
                               O.compute_barkspectrum(xm, sample_rate=SR, mode="mono").numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(PF.compute_rms_energy(xm).numpy(), O.compute_rms_energy(xm).numpy(), rtol=2e-6)
    np.testing.assert_allclose(PF.compute_crest_factor(xm).numpy(), O.compute_crest_factor(xm).numpy(), rtol=0, atol=2e-5)
    for xx in (x, xm, torch.cat([x[:1], torch.zeros(1, 2, x.shape[-1])])):  # the last one has an all-silent item (NaN frames)
        np.testing.assert_allclose(PF.compute_spectral_centroid(xx, SR).numpy(), O.compute_spectral_centroid(xx, SR).numpy(),
                                   rtol=0, atol=2e-5)
    lufs = PF.compute_lufs(x, SR)
    assert lufs.shape == (3, 1) and bool(((lufs > -30) & (lufs < 5)).all())
    with pytest.raises(NotImplementedError):
        PF.compute_barkspectrum(x, 48000, mode="mono")   # the reference's positional-argument slip (utils.py:88)
    from st_ito.utils import get_mir_feature_embeds, load_mir_feature_extractor
    feats = get_mir_feature_embeds(x, load_mir_feature_extractor(), SR)
    assert {k: tuple(v.shape) for k, v in feats.items()} == {"lufs": (3, 1), "rms": (3, 2), "crest": (3, 2),
                                                             "barkspectrum": (3, 24), "spectral_centroid": (3, 20)}


def test_lufs_on_gpu_vs_host_bs1770(dev):
    """compute_lufs (features.py:267-299) on the GPU (stito_lufs: per-sample cross-channel normalisation, K-weighting through
    the float64 biquad cascade, gated 400 ms block energies) against the host restatement of pyloudnorm's meter
    (st_ito.loudness.integrated_loudness) on the same normalised signal: stereo and mono, lengths whose block edges do and
    do not fall on multiples of the hop, a quiet tail that the relative gate removes, 44.1 kHz, and digital silence."""
    from st_ito import features as Fx
    from st_ito.loudness import integrated_loudness
    cases = []
    for seed, chs, n, sr in ((1, 2, 480000, 48000), (2, 1, 100001, 48000), (3, 2, 96000, 48000), (4, 2, 200000, 44100), (5, 2, 30000, 48000)):
        x = O.synth_audio(seed, chs, n)
        if seed == 1:
            x[:, n // 2:] *= 1e-3          # the second half falls under the relative gate
        cases.append((x, sr))
    for x, sr in cases:
        xb = torch.stack([x, 0.25 * x.flip(-1)])
        got = Fx.compute_lufs(xb.to(dev), sr).cpu().numpy().reshape(-1)
        for b in range(2):
            xx = xb[b]
            pk = xx.abs().max(dim=0)[0].clamp(min=1e-8)
            xn = (xx / pk[None]).repeat(2 // xx.shape[0], 1) if xx.shape[0] == 1 else xx / pk[None]
            ref = integrated_loudness(xn.permute(1, 0).numpy(), sr)
            assert abs(got[b] - ref) < 1e-3, (got[b], ref, xx.shape, sr)
    z = torch.zeros((1, 2, 48000))
    assert Fx.compute_lufs(z.to(dev), 48000).item() == float("-inf")   # 0 / clamp -> silence -> every block under the absolute gate
    with pytest.raises(ValueError):
        Fx.compute_lufs(torch.zeros((1, 2, 1000)).to(dev), 48000)


@pytest.mark.parametrize("chs,n,sr", [(1, 48000, 48000), (2, 70001, 48000), (2, 30000, 44100)])
def test_chorus_vs_oracle(dev, chs, n, sr):
    """BasicChorus (effects.py:962-985; STITO_FX_CHORUS, csrc/modfx.hip) against the oracle's restatement of
    juce::dsp::Chorus: default parameters, the extremes (depth 1, feedback 1: the feedback path through the interpolated
    delay line; centre delay at its 1 ms clamp), a population with different parameters per candidate, and the stage inside
    a chain (EQ -> chorus -> gain).  The delay d = lfo * fs / 1000 is a float32 near 300 .. 1400 samples (ulp 3e-5 .. 1.2e-4): one
    ulp of sin() in the LFO moves the output by ulp(d) * |v[n - 1] - v[n]| ~ 1e-5 .. 1e-4 of its peak, and no two sine
    implementations agree in every last bit (the GPU's table is the correctly rounded sine; glibc's sinf is within 0.56 ulp).
    So every case is checked twice: against the oracle walking its own libm oscillator (bar 2e-4 of the peak: the inherent
    spread) and against the oracle fed the GPU's LFO table, where everything else -- delay line, interpolation, feedback,
    mix -- must agree to 2e-6."""
    from st_ito import effects as E
    from st_ito.engine import render_population, compile_chain
    x = 0.5 * O.synth_audio(40 + chs, chs, n).numpy()
    settings = [None, dict(centre_delay_ms=0.1, depth=1.0, feedback=0.95, mix=1.0), dict(centre_delay_ms=20.0, depth=1.0, feedback=0.3, mix=0.7),
                dict(centre_delay_ms=3.3, depth=0.0, feedback=0.0, mix=0.5)]
    for st_ in settings:
        oc, pc = O.OracleChorus(), E.BasicChorus()
        for k, v in (st_ or {}).items():
            oc.parameters[k].set_value(v); pc.parameters[k].set_value(v)
        got = pc.process(x, sr)
        ref = oc.process(x, sr)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9)
        assert err < 2e-4, (st_, err)
        oc.lfo_table = E.chorus_lfo_device(sr, n, dev).cpu().numpy()
        ref = oc.process(x, sr)
        err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-9)
        assert err < 2e-6, (st_, err)
    # population: one launch, per-candidate parameters
    W = np.random.default_rng(9).random((5, 5))
    pl = {"Chorus": {"class_path": E.BasicChorus, "instance": E.BasicChorus(), "num_channels": 1, "fixed_parameters": {},
                     "parameter_names": ["rate_hz", "centre_delay_ms", "depth", "feedback", "mix"], "num_params": 5}}
    audio, _ = render_population(pl, torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), sr)
    op = O.make_plugins(["Chorus"])
    op["Chorus"]["instance"].lfo_table = E.chorus_lfo_device(sr, n, dev).cpu().numpy()
    for p_ in range(5):
        for name, v in zip(op["Chorus"]["parameter_names"], W[p_]):
            op["Chorus"]["instance"].parameters[name].raw_value = v
        ref = op["Chorus"]["instance"].process(x, sr)
        assert np.abs(audio[p_].cpu().numpy() - ref).max() < 2e-6 * max(np.abs(ref).max(), 1.0)
    # inside a chain, through process_audio
    from st_ito.style_transfer import process_audio
    chain = [("ParametricEQ", E.BasicParametricEQ, 1), ("Chorus", E.BasicChorus, 1), ("Gain", E.BasicGain, 1)]
    pp = E.make_plugins(chain)
    oo = O.make_plugins(["ParametricEQ", "Chorus", "Gain"])
    oo["Chorus"]["instance"].lfo_table = E.chorus_lfo_device(sr, n, dev).cpu().numpy()
    w = np.random.default_rng(10).random(18 + 5 + 1)
    got = process_audio(x, w, sr, pp)
    ref = O.process_audio(x, w, sr, oo)
    assert np.abs(got - ref).max() < 5e-6


def test_dasp_compressor_vs_oracle(dev):
    """apply_random_compressor (dsp.py:49-78 -> dasp_pytorch.functional.compressor; stito_dasp_compressor) against the
    oracle's restatement, which evaluates the smoothing filter by frequency sampling like the library: mono and stereo
    (side chain = channel sum), a batch, and the wrapper's own draw of the threshold.  The bar (2e-4 of the peak) is the
    float32 FFT noise of the library's frequency-sampled filter itself (gains of up to 30 dB through a 2^17-point float32
    transform: measured 6e-5 against the float64 recursion the GPU runs; tests/test_oracle_golden.py has that comparison)."""
    from st_ito import _hip, dsp as D
    L = _hip.lib()
    for chs, n in ((1, 48000), (2, 100000)):
        x = torch.stack([O.synth_audio(60 + i, chs, n) * (0.9 if i else 0.2) for i in range(2)])
        for thr in (-40.0, -6.0):
            ref = O.dasp_compressor(x, 48000, thr)
            xd = x.to(dev).contiguous()
            out = torch.empty_like(xd)
            _hip.check(L.stito_dasp_compressor(_hip.ptr(xd), 2, chs, n, 48000.0, thr, 4.0, 100.0, 24.0, 0.0, _hip.ptr(out), _hip.stream_ptr()))
            err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 2e-4, (chs, thr, err)
    np.random.seed(4)
    y, thr = D.apply_random_compressor(x[0], 48000)
    np.random.seed(4)
    assert thr == np.random.uniform(-48, 0) and y.shape == x[0].shape
    assert (y - O.dasp_compressor(x[:1], 48000, float(np.float32(thr)))[0]).abs().max().item() < 2e-4


def test_dsp_module_random_effects(dev):
    """st_ito/dsp.py mirror: random-parameter distortion, noise reverb and compressor on the GPU, loudness on the host."""
    from st_ito import dsp as PD
    from st_ito.loudness import integrated_loudness
    x = O.synth_audio(81, 2, 60000)
    np.random.seed(3)
    y, drive = PD.apply_random_simple_distortion(x, SR)
    assert 0 <= drive <= 32 and y.shape == x.shape
    np.testing.assert_allclose(y.numpy(), np.tanh(x.numpy() * np.float32(10 ** (np.float32(drive) / 20))), rtol=0, atol=2e-6)
    np.random.seed(4)
    y, mix = PD.apply_random_reverb(x, SR)
    rv = O.OracleNoiseShapedReverb(sample_rate=SR, seed=0)
    for b, d in enumerate([0.6, 0.4, 0.4, 0.5, 0.2, 0.3, 0.3, 0.2, 0.1, 0.1, 0.2, 0.1]):
        rv.parameters[f"band{b}_decay"].set_value(float(np.float32(d)))
        rv.parameters[f"band{b}_gain"].set_value(1.0)
    rv.parameters["mix"].set_value(float(np.float32(mix)))
    ref = rv.process(x.numpy(), SR)
    assert np.abs(y.numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    yb, _ = PD.apply_random_reverb(x[None], SR)
    assert yb.shape == (1, 2, 60000)
    z = PD.normalize_loudness(x, SR, -23.0)
    assert abs(integrated_loudness(z.numpy().T, SR) - (-23.0)) < 0.05
    yc, thr = PD.apply_random_compressor(x, SR)          # (against the oracle: test_dasp_compressor_vs_oracle)
    assert -48 <= thr <= 0 and yc.shape == x.shape and torch.isfinite(yc).all() and yc.abs().max() <= x.abs().max() + 1e-6


def test_mfcc_feature_embeds_vs_oracle(dev):
    """get_mfcc_feature_embeds (utils.py:116-159; torchaudio MFCC restated, unpinned): the STFT/mel/log part
    runs through stito_logmel (no-centre mode, HTK mel bands), DCT and statistics in stito_mfcc_stats."""
    from st_ito.utils import get_mfcc_feature_embeds, load_mfcc_feature_extractor
    model = load_mfcc_feature_extractor()
    assert model.embed_dim == 75
    x = torch.stack([O.synth_audio(91, 2, 80000), 0.05 * O.synth_audio(92, 2, 80000)])
    x[1, :, 40000:] = 0.0   # half of item 1 is digital silence: exercises the 1e-10 clamp and the 80 dB floor
    for midside in (False, True):
        got = get_mfcc_feature_embeds(x, model, SR, midside=midside)["mono"]
        ref = O.mfcc_feature_embeds(x, SR, midside=midside)
        assert got.shape == ref.shape == (2, 150 if midside else 75) and got.device == x.device
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(np.linalg.norm(got.numpy(), axis=1), 1.0, atol=1e-6)
    got = get_mfcc_feature_embeds(x, model, 44100)["mono"]       # utils.py:130-131: resampled to 48 kHz first
    ref = O.mfcc_feature_embeds(O.resample_sinc(x, 44100, 48000), SR)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("kinds,chs", [(["ParametricEQ", "Gain", "Reverb"], 2), (["ParametricEQ", "Gain"], 1),
                                        (["Gain", "ParametricEQ"], 2), (["ParametricEQ", "Gain", "ParametricEQ", "Gain"], 2)])
def test_eq_store_fusion_paths(dev, kinds, chs):
    """The EQ kernel's store absorbs a following Gain stage and, at the end of the chain, the peak scan
    (dsp.hip PostOp).  Chains where the fused pair is in the middle, at the end, absent, and repeated; with and
    without per-stage normalisation (which disables the fusion) -- all against the oracle's stage-by-stage path."""
    from st_ito.style_transfer import process_audio
    op, pp = _plugins_pair(kinds)
    D = sum(p["num_params"] for p in op.values())
    x = O.synth_audio(33, chs, 20011).numpy()
    W = np.random.default_rng(len(kinds) * 7 + chs).random((3, D))
    got, peaks = _render_gpu(pp, x, W, dev, normalize=False)
    for p in range(3):
        ref = _oracle_chain_raw(op, x, W[p])
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(got[p] - ref).max() / scale < 2e-5
        assert peaks[p] == np.abs(got[p]).max()          # fused or separate, the peak is the exact max of what was stored
    for ns in (False, True):
        ref = O.process_audio(x.copy(), W[0], SR, op, normalize_stages=ns)
        out = process_audio(x.copy(), W[0], SR, pp, normalize_stages=ns)
        np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5)
