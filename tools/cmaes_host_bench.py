#!/usr/bin/env python
"""Host cost of the replicated CMA-ES step (ask / tell of the WHOLE population on every rank) with the BLAS limited to one thread
(st_ito/cmaes.py) and as configured, per population size: what does not shrink with the rank count in a multi-GPU run.
    python tools/cmaes_host_bench.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/st-ito_amd")
import numpy as np
from st_ito import cmaes
t0 = time.perf_counter()
with cmaes._one_blas_thread(): pass
print(f"first limit (controller creation): {1e3*(time.perf_counter()-t0):.2f} ms")
t0 = time.perf_counter()
for _ in range(1000):
    with cmaes._one_blas_thread(): pass
print(f"enter + exit: {1e3*(time.perf_counter()-t0)/1000:.4f} ms per call")
from threadpoolctl import threadpool_info
print([(d['internal_api'], d['num_threads'], d.get('threading_layer')) for d in threadpool_info()])
import contextlib
for P in (32, 256, 512, 2048):
    for limited in (True, False):
        if not limited:
            cmaes._one_blas_thread = lambda: contextlib.nullcontext()
        es = cmaes.CMAEvolutionStrategy(np.ones(45)*0.5, 0.33, {"bounds":[0,1],"popsize":P,"seed":42})
        rng=np.random.default_rng(0); ta=tt=0
        for it in range(22):
            t0=time.perf_counter(); W=es.ask(); t1=time.perf_counter(); es.prefetch()
            f=list(rng.random(P)); t2=time.perf_counter(); es.tell(W,f); t3=time.perf_counter()
            if it>=2: ta+=t1-t0; tt+=t3-t2
        print(f"lambda {P:5d} {'one BLAS thread' if limited else 'BLAS as configured':18s}: ask {ta/20*1e3:.3f} ms  tell {tt/20*1e3:.3f} ms")
    import importlib; importlib.reload(cmaes)
