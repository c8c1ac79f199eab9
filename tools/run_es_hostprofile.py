"""Host-side profile of one whole run_es call at the reference's CLI-default point (pop 32, basic chain, find_w0): where do the
milliseconds go that are not kernels?   python tools/run_es_hostprofile.py"""
import cProfile, io, os, pstats, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import synth_audio
from st_ito import effects as E
from st_ito.style_transfer import load_plugins, run_es, process_audio
from st_ito.utils import get_param_embeds, make_synthetic_param_model
SR = 48000
model = make_synthetic_param_model(seed=0, input_norm="minmax")
with contextlib.redirect_stdout(io.StringIO()):
    plugins, D, _ = load_plugins(E.make_plugins("basic"))
n = int(5.0 * SR)
x = synth_audio(300, 2, n)[None]
tg = torch.from_numpy(process_audio(synth_audio(301, 2, n).numpy(), np.random.default_rng(3).random(D), SR, plugins))[None]
kw = dict(max_iters=24, popsize=32, sigma0=0.33, random_crop=False, find_w0=True, seed=11, early_stop=False)
with contextlib.redirect_stdout(io.StringIO()):
    run_es(x.clone(), tg.clone(), SR, plugins, model, get_param_embeds, **dict(kw, max_iters=2))
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        r = run_es(x.clone(), tg.clone(), SR, plugins, model, get_param_embeds, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"run_es: {dt * 1e3:.1f} ms for {len(r['fval_history'])} iterations + find_w0 = {32 * len(r['fval_history']) / dt:.0f} cand/s")
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); r = run_es(x.clone(), tg.clone(), SR, plugins, model, get_param_embeds, **kw); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
