"""GPU: the BASELINE.json configs that bench.py does not run, at the shape one GPU sees, against the
oracle -- configs[2] (16 pairs x pop 128, stereo 10 s), configs[4]'s per-GPU share (pop 128, stereo
30 s, 96 000-tap convolution reverb in the chain) -- and evaluate(random_crop=True), the setting of
the reference's PST benchmark (scripts/eval/eval_pst.py:974-991; crop rule style_transfer.py:505-518).
Tolerance: losses within 1e-4 relative (north_star)."""
import numpy as np
import pytest
import torch

import st_ito_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000
CHAIN5 = ["ParametricEQ", "Compressor", "Reverb", "ParametricEQ", "Gain"]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito import _hip
    _hip.lib()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def models(dev):
    from st_ito.models.panns import Cnn14
    om = O.make_synthetic_model(0)
    pm = Cnn14(512, SR, 2048, 1024, 128, 20, 20000, True, "minmax")
    pm.load_state_dict(om.state_dict())
    return om, pm.eval().to(dev)


def _close(a, b, rel=1e-4):
    return abs(a - b) <= rel * max(1.0, abs(b))


def test_config2_full_shape_16_pairs_x_128_vs_oracle(dev, models):
    """configs[2] at full shape: 16 (input, target) pairs x pop 128, 48 kHz stereo 10 s, 5-effect chain, one
    PopulationEvaluator pass over the 2 048 candidates.  Oracle: two candidates of each of four pairs
    (first / middle / last pair, first and last candidate) through O.evaluate; properties at full size:
    all losses finite cosines, and a pair's candidates evaluated alone (single-pair evaluator) are bitwise
    the ones evaluated inside the 16-pair batch."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds
    om, pm = models
    B, P, n, D = 16, 128, 480000, 45
    op = O.make_plugins(CHAIN5)
    xs = torch.stack([O.synth_audio(300 + b, 2, n) for b in range(B)])
    ts = torch.stack([O.synth_audio(400 + b, 2, n) * (0.25 + 0.05 * b) for b in range(B)])
    W = np.random.default_rng(16).random((B * P, D))
    te = get_param_embeds(ts.clone(), pm, SR)
    assert te["mid"].shape == (B, 512)
    ev = PopulationEvaluator(xs, SR, E.make_plugins("bench5"), pm, te)
    loss, emb, _ = ev.evaluate(W)
    lossn = loss.cpu().numpy()
    assert lossn.shape == (B * P,) and np.isfinite(lossn).all() and (np.abs(lossn) <= 1.0001).all()
    for b in (0, 5, 10, 15):
        te_o = O.get_param_embeds(ts[b:b + 1].clone(), om, SR)
        for k in ("mid", "side"):
            assert (te[k][b].cpu() - te_o[k][0]).abs().max() / te_o[k].abs().max() < 1e-4
        idx = [b * P, b * P + P - 1]
        f_ref, e_ref, _ = O.evaluate([W[i] for i in idx], xs[b:b + 1], SR, op, te_o, om)
        for j, i in enumerate(idx):
            assert _close(lossn[i], f_ref[j]), (b, i, lossn[i], f_ref[j])
            for k in ("mid", "side"):
                rel = (emb[k][i].cpu() - e_ref[k][j]).abs().max() / e_ref[k][j].abs().max()
                assert rel < 1e-4, (b, i, k, float(rel))
    # batch position / pair count independence, bitwise
    for b in (3, 15):
        ev1 = PopulationEvaluator(xs[b:b + 1], SR, E.make_plugins("bench5"), pm, {k: v[b:b + 1] for k, v in te.items()})
        l1, e1, _ = ev1.evaluate(W[b * P:(b + 1) * P])
        assert torch.equal(l1, loss[b * P:(b + 1) * P])
        assert torch.equal(e1["mid"], emb["mid"][b * P:(b + 1) * P])


def test_config4_share_96000_tap_conv_reverb_30s_vs_oracle(dev, models):
    """configs[4] per-GPU share at full shape: pop 128, 48 kHz stereo 30 s, chain EQ / compressor / noise-shaped
    convolution reverb with a 96 000-tap (2 s) IR / EQ / gain.  One candidate against the oracle (float64
    FFT convolution): rendered audio within 2e-5 of peak, loss within 1e-4; all 128 losses finite; the
    candidate evaluated alone is bitwise the one inside the batch."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator
    from st_ito.utils import get_param_embeds
    om, pm = models
    P, n, taps = 128, 1440000, 96000
    bank = O.make_noise_bank(taps, 1023, SR, seed=11)

    def chain(eq, comp, rv, gn):
        pl = {}
        for name, inst, nch in (("ParametricEQ", eq(), 1), ("Compressor", comp(), 1), ("ConvReverb", rv(noise_bank=bank), 2),
                                ("ParametricEQ2", eq(), 1), ("Gain", gn(), 1)):
            names = list(inst.parameters.keys())
            pl[name] = {"class_path": type(inst), "num_params": len(names), "num_channels": nch, "fixed_parameters": {},
                        "instance": inst, "parameter_names": names}
        return pl
    op = chain(O.OracleParametricEQ, O.OracleCompressor, O.OracleNoiseShapedReverb, O.OracleGain)
    pp = chain(E.BasicParametricEQ, E.BasicCompressor, E.NoiseShapedReverb, E.BasicGain)
    D = 18 + 4 + 25 + 18 + 1
    x = O.synth_audio(91, 2, n)[None]
    tgt = O.synth_audio(92, 2, n)[None]
    W = np.random.default_rng(44).random((P, D))
    te = get_param_embeds(tgt.clone(), pm, SR)
    ev = PopulationEvaluator(x, SR, pp, pm, te)
    loss, emb, _ = ev.evaluate(W)
    lossn = loss.cpu().numpy()
    assert np.isfinite(lossn).all() and (np.abs(lossn) <= 1.0001).all()
    te_o = O.get_param_embeds(tgt.clone(), om, SR)
    f_ref, _, a_ref = O.evaluate([W[5]], x, SR, op, te_o, om)
    assert _close(lossn[5], f_ref[0]), (lossn[5], f_ref[0])
    l1, _, a1 = ev.evaluate(W[5:6], want_audio=True)
    assert l1.item() == lossn[5]
    err = (a1[0].cpu() - a_ref[0]).abs().max().item()   # both peak-normalised: peak = 1
    assert err < 2e-5, err


@pytest.mark.parametrize("extra", [8000, 100000])
def test_evaluate_random_crop_vs_oracle(dev, models, extra):
    """random_crop=True (style_transfer.py:505-518): with 0 < L - 262144 <= 16384 the reference evaluates
    x[..., 0:262144]; beyond that one start index in [16384, L - 262144) is drawn from np.random for the whole
    population.  Same seeded RandomState on both sides -> same crop; losses within 1e-4."""
    from st_ito import effects as E
    from st_ito.engine import PopulationEvaluator, CROP_LEN
    from st_ito.utils import get_param_embeds
    om, pm = models
    n = CROP_LEN + extra
    x = O.synth_audio(55, 2, n)[None]
    tgt = O.synth_audio(56, 2, CROP_LEN)[None]
    op = O.make_plugins(CHAIN5)
    W = np.random.default_rng(extra).random((3, 45))
    te = get_param_embeds(tgt.clone(), pm, SR)
    te_o = O.get_param_embeds(tgt.clone(), om, SR)
    ev = PopulationEvaluator(x, SR, E.make_plugins("bench5"), pm, te)
    loss, _, audio = ev.evaluate(W, random_crop=True, rng=np.random.RandomState(9), want_audio=True)
    f_ref, _, a_ref = O.evaluate(list(W), x, SR, op, te_o, om, random_crop=True, rng=np.random.RandomState(9))
    assert audio.shape[-1] == CROP_LEN and a_ref.shape[-1] == CROP_LEN     # the crop happens in both regimes
    np.testing.assert_allclose(loss.cpu().numpy(), np.array(f_ref), rtol=1e-4, atol=5e-6)
    assert (audio.cpu() - a_ref).abs().max().item() < 5e-5
    if extra <= 16384:  # start index 0: nothing drawn
        l0, _, _ = ev.evaluate(W, random_crop=True, rng=None)
        assert torch.equal(l0, loss)
    # without random_crop the whole (longer) input is used
    lf, _, af = ev.evaluate(W[:1], random_crop=False, want_audio=True)
    assert af.shape[-1] == n
