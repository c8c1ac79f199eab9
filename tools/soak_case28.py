import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, os.path.join(ROOT, "st-ito_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import st_ito_oracle as O
from st_ito import effects as E, engine
import soak
rng = np.random.default_rng(0)
for case in range(29):
    c = soak.draw_case(rng, case)
print(soak.describe(c))
dev = torch.device("cuda", 0)
# compressor alone on the output of the two EQs (taken from the ORACLE, so both sides see the same input)
kinds = c["kinds"]
op2 = O.make_plugins(kinds[:2], c["with_bypass"])
D2 = sum(p["num_params"] for p in op2.values())
opc = O.make_plugins(["Compressor"], c["with_bypass"])
ppc = E.make_plugins([("Compressor", E.BasicCompressor, 1)], c["with_bypass"])
for k in ("fixed_parameters",):
    opc["Compressor"][k] = dict(c["op"]["Compressor"][k]); ppc["Compressor"][k] = dict(c["pp"]["Compressor"][k])
for p in range(c["P"]):
    w = c["W"][p]
    mid = O.process_audio(c["x"].copy(), w[:D2], 48000, op2, normalize_stages=True)
    wc = w[D2:]
    ref = O.process_audio(mid.copy(), wc, 48000, opc, normalize_stages=True)
    a, pk = engine.render_population(ppc, torch.from_numpy(mid).to(dev), torch.from_numpy(wc[None]).to(dev), 48000, chain=engine.compile_chain(ppc, True))
    engine.normalize_audio_(a, pk)
    got = a[0].cpu().numpy()
    inst = opc["Compressor"]["instance"]
    print(p, "in", mid.ravel(), "\n  ref", ref.ravel(), "\n  got", got.ravel(), "\n  params", {k: v.get_value() if hasattr(v, "get_value") else v for k, v in inst.parameters.items()})
    a2, pk2 = engine.render_population(ppc, torch.from_numpy(mid).to(dev), torch.from_numpy(wc[None]).to(dev), 48000, chain=engine.compile_chain(ppc, False))
    ref2 = O.process_audio(mid.copy(), wc, 48000, opc, normalize_stages=False) 
    print("  un-normalised ref", (ref2 * 1.0).ravel(), "got", (a2[0] / pk2[0].clamp(min=1e-8)).cpu().numpy().ravel())
