"""Why do product and oracle select different vectors in the compressor case-study point?  Records every tell()'s fitness on both sides."""
import os, sys
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("st-ito_amd", "oracle", os.path.join("st-ito_amd", "scripts")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import st_ito_oracle as O
import eval_case_study as C
from st_ito import cmaes
from st_ito.models.panns import Cnn14

dev = torch.device("cuda", 0)
om = O.make_synthetic_model(0)
pm = Cnn14(512, 48000, 2048, 1024, 128, 20, 20000, True, "minmax"); pm.load_state_dict(om.state_dict()); pm = pm.eval().to(dev)
src = [O.synth_audio(801, 2, C.MIN_LEN + 30000), O.synth_audio(802, 1, C.MIN_LEN + 50000)]
log = {"hip": [], "cpu": []}
def factory(tag):
    class ES(cmaes.CMAEvolutionStrategy):
        def tell(self, W, f):
            log[tag].append((np.asarray(W).copy(), np.asarray(f, dtype=np.float64).copy()))
            return super().tell(W, f)
    return ES
plugin_name, kind, value = "pb_Compressor", "Compressor", 0.3
spec, param, lo, hi = C.get_case(plugin_name)
import st_ito.style_transfer as ST
orig = ST.cma.CMAEvolutionStrategy
ST.cma.CMAEvolutionStrategy = factory("hip")
got = C.study_point(spec, plugin_name, param, value, lambda r: (src[0], src[1]), pm, np.random.RandomState(11), max_iters=2, popsize=6, seed=4)
ST.cma.CMAEvolutionStrategy = orig
op = O.make_plugins([kind], with_bypass=True)
op = OrderedDict([(plugin_name, op[kind])])
op[plugin_name]["fixed_parameters"] = dict(spec[plugin_name]["fixed_parameters"])
est, fopt, tv, wopt = O.case_study_point(op, plugin_name, param, value, src[0], src[1], om, factory("cpu"), np.random.RandomState(11), max_iters=2, popsize=6, seed=4)
for it, (a, b) in enumerate(zip(log["hip"], log["cpu"])):
    print("iteration", it, "same W:", np.array_equal(a[0], b[0]))
    print("  threshold slot:", np.round(a[0][:, 1], 4))
    print("  hip fitness:", a[1]); print("  cpu fitness:", b[1]); print("  diff:", a[1] - b[1])
print("hip wopt", got["wopt"], got["fopt"]); print("cpu wopt", wopt, fopt)
