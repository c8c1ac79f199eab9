"""GPU: the randomised soak inside the suite: seeded random chains of three to five stages (repeats,
Distortion / Delay / Reverb / Compressor / EQ / Gain in any order), ragged lengths (1, 5, 191, 193, 4096, 4097, 30011, 48000,
65536, 100003 samples), mono / stereo, bypass slots, fixed parameters and per-stage normalisation, rendered through the C ABI and
held against the oracle candidate by candidate; and random whole evaluate() calls (chain -> log-mel -> Cnn14 -> loss, three
input_norm modes) against oracle.evaluate.  The generators are tools/soak.py / tools/soak_eval.py (longer hunts: profiles/)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from st_ito import _hip
    _hip.lib()
    return torch.device("cuda", 0)


def test_random_chains_of_three_or_more_stages_vs_oracle(dev):
    """40 seeded cases, every chain >= 3 stages.  Bar per case: 1e-4 of the peak (north_star's tolerance; the single-effect and
    bench-chain tests hold 2e-6 ... 2e-5).  What sits between those bars is conditioning, not a kernel: a Distortion multiplies
    a difference by up to 10^(48/20) = 251 at a zero crossing and a compressor behind another stage turns a float32-rounding
    difference of its input level into a gain difference (tools/soak_case.py replays a case prefix by prefix)."""
    import soak
    rng = np.random.default_rng(6)
    rows, worst = [], 0.0
    for case in range(40):
        c = soak.draw_case(rng, case, min_fx=3)
        assert len(c["kinds"]) >= 3
        err = soak.render_case(c, dev)
        rows.append(f"case {case:2d}: {soak.describe(c)}  {err:.2e}")
        worst = max(worst, err)
    print("\n".join(rows))
    print(f"worst of 40: {worst:.2e} of peak")
    bad = [r for r in rows if float(r.rsplit(None, 1)[1]) > 1e-4]
    assert not bad, bad


def test_random_evaluate_calls_vs_oracle(dev):
    """8 seeded evaluate() cases: per-candidate loss within 1e-6 of the oracle's (measured ~1e-7: the loss is a cosine of
    L2-normalised 512-vectors, float32 rounding)."""
    import soak_eval
    rng = np.random.default_rng(7)
    worst = 0.0
    for case in range(8):
        c = soak_eval.draw_eval_case(rng, case)
        err = soak_eval.eval_case(c, rng, dev)
        print(f"case {case}: {'+'.join(c['kinds'])} norm={c['norm']} chs={c['chs']} n={c['n']} P={c['P']}: |dloss| {err:.2e}")
        assert err <= 1e-6, (c, err)
        worst = max(worst, err)
    print(f"worst of 8: {worst:.2e}")
