"""Seeded (mu/mu_w, lambda)-CMA-ES with box constraints, the optimiser behind run_es.

The reference drives pycma (`cma.CMAEvolutionStrategy(w0, sigma0, {"bounds": [0, 1],
"popsize": P})`, st_ito/style_transfer.py:614, 624, 651-652, 672-673); pycma is an un-pinned
third-party package that is not available here, so this module implements the same interface
(ask / tell / result / disp / stop) with the algorithm pycma runs by default (Hansen, "The CMA
Evolution Strategy: A Tutorial", 2016/2023 revision): weighted recombination, CSA step-size control
with c_sigma = (mu_eff + 2) / (N + mu_eff + 3), rank-one + rank-mu covariance update with ACTIVE
(negative) recombination weights (pycma `CMA_active=True`), lazy eigendecomposition, and pycma's
BoxConstraintsLinQuadTransformation for the [0, 1] bounds -- the initial mean is the transformation's
INVERSE image of x0, so the first distribution is centred on w0 itself.  It is deterministic under
`seed`, uses only fitness ranks plus the best value, and every rank of a multi-GPU run steps an
identical replica.  Not bit-compatible with pycma (different random stream); same update rule.
"""
from __future__ import annotations

import contextlib
import math
from typing import List, Optional, Sequence

import numpy as np

# ask() and tell() are a handful of (lambda x N) x (N x N) products with N <= 50: a few MFLOP.  A multi-threaded BLAS turns them into
# a liability as soon as lambda crosses its threading threshold -- measured with OpenBLAS on 8 cores: lambda 512 ask 11.7 ms / tell
# 11.6 ms against 0.6 / 0.9 ms on one thread, lambda 2048 (8 GPUs x 256) 7.9 / 13.5 against 2.3 / 1.7 -- and every rank of a
# multi-GPU run steps its own replica on the same host, so eight ranks would each wake a full thread pool.  Both calls therefore run
# with the BLAS limited to one thread (threadpoolctl, when it is installed); a single thread also makes the replicas' arithmetic
# independent of the host's core count.
_BLAS = None


def _one_blas_thread():
    global _BLAS
    if _BLAS is None:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS = ThreadpoolController()
        except Exception:  # noqa: BLE001 -- not installed: run as the BLAS is configured
            _BLAS = False
    if _BLAS:
        try:
            return _BLAS.limit(limits=1, user_api="blas")
        except Exception:  # noqa: BLE001 -- a BLAS build the controller cannot steer: run as configured from now on
            _BLAS = False
    return contextlib.nullcontext()


class BoundTransform:
    """pycma BoxConstraintsLinQuadTransformation for scalar bounds [lb, ub]: identity on
    [lb+al, ub-au], quadratic towards the bounds, periodic mirroring outside [lb-al, ub+au]."""

    def __init__(self, lb: float, ub: float):
        self.lb, self.ub = float(lb), float(ub)
        self.al = min((self.ub - self.lb) / 2.0, (1.0 + abs(self.lb)) / 20.0)
        self.au = min((self.ub - self.lb) / 2.0, (1.0 + abs(self.ub)) / 20.0)

    def __call__(self, y: np.ndarray) -> np.ndarray:
        lb, ub, al, au = self.lb, self.ub, self.al, self.au
        y = np.array(y, dtype=np.float64, copy=True)
        lo, hi = lb - al, ub + au
        # pycma's shift_or_mirror_into_invertible: values inside [lo, hi] -- every sample of a run that behaves -- are left
        # bit for bit as they are; only the others are shifted by whole periods into [lo, lo + period) and the upper half
        # mirrored back
        out = (y < lo) | (y > hi)
        if out.any():
            period = 2.0 * (hi - lo)
            v = lo + np.mod(y[out] - lo, period)
            y[out] = np.where(v > hi, 2.0 * hi - v, v)
        # (both quadratic branches evaluated everywhere and selected: the same arithmetic per element as assigning through
        # boolean masks, a third of the time for a 256 x 45 population -- this runs on the host between two GPU passes)
        return np.where(y < lb + al, lb + (y - (lb - al)) ** 2 / (4.0 * al),
                        np.where(y > ub - au, ub - (y - (ub + au)) ** 2 / (4.0 * au), y))

    def inverse(self, x: np.ndarray) -> np.ndarray:
        """Genotype in [lb - al, ub + au] whose image is x (x is clipped into [lb, ub] first), as pycma
        maps x0 before the first ask(): identity in the linear region, y = lb - al + 2 sqrt(al (x - lb))
        towards the lower bound and symmetrically towards the upper one."""
        lb, ub, al, au = self.lb, self.ub, self.al, self.au
        x = np.clip(np.asarray(x, dtype=np.float64), lb, ub)
        y = x.copy()
        low = x < lb + al
        y[low] = (lb - al) + 2.0 * np.sqrt(al * (x[low] - lb))
        up = x > ub - au
        y[up] = (ub + au) - 2.0 * np.sqrt(au * (ub - x[up]))
        return y


class _Result(tuple):
    """es.result: indexable like pycma's (xbest, fbest, evals_best, evaluations, iterations, xmean, stds)."""

    xbest = property(lambda s: s[0])
    fbest = property(lambda s: s[1])


class CMAEvolutionStrategy:
    def __init__(self, x0: Sequence[float], sigma0: float, inopts: Optional[dict] = None):
        opts = dict(inopts or {})
        self.N = N = len(x0)
        self.lam = int(opts.get("popsize", 4 + int(3 * math.log(N))))
        if self.lam < 2:
            raise ValueError("popsize must be >= 2")
        seed = opts.get("seed", None)
        self.rng = np.random.Generator(np.random.PCG64(seed))
        b = opts.get("bounds", None)
        self.boundary = BoundTransform(b[0], b[1]) if b is not None and b[0] is not None else None
        self.mean = np.asarray(x0, dtype=np.float64).copy()
        if self.boundary is not None:
            self.mean = self.boundary.inverse(self.mean)
        self.sigma = float(sigma0)
        self.sigma0 = float(sigma0)
        self.active = bool(opts.get("CMA_active", True))
        # recombination weights, Tutorial eq. (49)-(53): w'_i = ln((lam + 1) / 2) - ln i for all lam ranks
        self.mu = self.lam // 2
        wp = math.log((self.lam + 1) / 2.0) - np.log(np.arange(1, self.lam + 1))
        pos, neg = wp[: self.mu], wp[self.mu:]
        self.mueff = float(pos.sum() ** 2 / np.sum(pos ** 2))
        mueff_neg = float(neg.sum() ** 2 / np.sum(neg ** 2)) if len(neg) and np.any(neg != 0) else 0.0
        self.cc = (4 + self.mueff / N) / (N + 4 + 2 * self.mueff / N)
        self.cs = (self.mueff + 2) / (N + self.mueff + 3)
        alpha_cov = 2.0
        self.c1 = alpha_cov / ((N + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, alpha_cov * (0.25 + self.mueff + 1 / self.mueff - 2) / ((N + 2) ** 2 + alpha_cov * self.mueff / 2))
        self.damps = 1 + 2 * max(0.0, math.sqrt((self.mueff - 1) / (N + 1)) - 1) + self.cs
        w = np.zeros(self.lam)
        w[: self.mu] = pos / pos.sum()
        if self.active and len(neg) and neg.sum() != 0:
            a_mu = 1 + self.c1 / self.cmu
            a_mueff = 1 + 2 * mueff_neg / (self.mueff + 2)
            a_posdef = (1 - self.c1 - self.cmu) / (N * self.cmu)
            w[self.mu:] = min(a_mu, a_mueff, a_posdef) * neg / (-neg.sum())   # sum of negative weights = -min(...)
        self.weights_all = w
        self.weights = w[: self.mu]
        self.chiN = math.sqrt(N) * (1 - 1.0 / (4 * N) + 1.0 / (21 * N * N))
        self.pc = np.zeros(N)
        self.ps = np.zeros(N)
        self.B = np.eye(N)
        self.D = np.ones(N)
        self.C = np.eye(N)
        self.invsqrtC = np.eye(N)
        self.eigeneval = 0
        self.counteval = 0
        self.countiter = 0
        self.best_x: Optional[np.ndarray] = None
        self.best_f = float("inf")
        self.best_evals = 0
        self._geno: Optional[np.ndarray] = None
        self._z_next: Optional[np.ndarray] = None
        self._stop = {}
        self.maxiter = opts.get("maxiter", None)

    # -- pycma-like surface ------------------------------------------------------------------
    @property
    def popsize(self):
        return self.lam

    def ask(self) -> List[np.ndarray]:
        """lambda candidate solutions (phenotypes, inside the bounds)."""
        with _one_blas_thread():
            return self._ask()

    def _ask(self) -> List[np.ndarray]:
        z = self._z_next if self._z_next is not None else self.rng.standard_normal((self.lam, self.N))
        self._z_next = None
        y = z * self.D[None, :] @ self.B.T
        self._geno = self.mean[None, :] + self.sigma * y
        ph = self._geno.copy() if self.boundary is None else self.boundary(self._geno)
        return list(ph)  # rows of a fresh array (boundary() copies; without bounds _geno is not handed out: see below)

    def prefetch(self):
        """Draw the NEXT generation's standard normals now (0.2 ms for 256 x 45 on the host): they do not depend on tell(),
        so a caller whose fitness evaluation runs asynchronously on the GPU calls this between launching it and waiting
        for it.  The generator is used by ask() only, in the same order: runs with and without prefetch() are identical."""
        if self._z_next is None:
            self._z_next = self.rng.standard_normal((self.lam, self.N))

    def tell(self, solutions: Sequence[np.ndarray], function_values: Sequence[float]):
        with _one_blas_thread():
            self._tell(solutions, function_values)

    def _tell(self, solutions: Sequence[np.ndarray], function_values: Sequence[float]):
        f = np.asarray(function_values, dtype=np.float64)
        if len(f) != self.lam or self._geno is None:
            raise ValueError("tell() needs the fitness of the lambda solutions of the last ask()")
        N = self.N
        self.counteval += self.lam
        self.countiter += 1
        order = np.argsort(f, kind="stable")
        if f[order[0]] < self.best_f:
            self.best_f = float(f[order[0]])
            self.best_x = np.asarray(solutions[order[0]], dtype=np.float64).copy()
            self.best_evals = self.counteval - self.lam + int(order[0]) + 1
        G = self._geno[order]
        X = G[: self.mu]
        old = self.mean
        self.mean = self.weights @ X
        ymean = (self.mean - old) / self.sigma
        self.ps = (1 - self.cs) * self.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * (self.invsqrtC @ ymean)
        hsig = (np.linalg.norm(self.ps) / math.sqrt(1 - (1 - self.cs) ** (2 * self.countiter)) / self.chiN) < (1.4 + 2 / (N + 1))
        self.pc = (1 - self.cc) * self.pc + (math.sqrt(self.cc * (2 - self.cc) * self.mueff) * ymean if hsig else 0.0)
        Y = (G - old[None, :]) / self.sigma
        wo = self.weights_all.copy()
        nz = wo < 0
        if nz.any():  # Tutorial eq. (46): negative weights scaled so that a negative update keeps C positive definite
            mah2 = np.sum((Y[nz] @ self.invsqrtC.T) ** 2, axis=1)
            wo[nz] = wo[nz] * N / np.maximum(mah2, 1e-300)
        dh = (0 if hsig else 1) * self.cc * (2 - self.cc)
        self.C = ((1 + self.c1 * dh - self.c1 - self.cmu * self.weights_all.sum()) * self.C + self.c1 * np.outer(self.pc, self.pc)
                  + self.cmu * (Y.T * wo[None, :]) @ Y)
        self.sigma *= math.exp(min(1.0, (self.cs / self.damps) * (np.linalg.norm(self.ps) / self.chiN - 1)))
        if self.counteval - self.eigeneval > self.lam / (self.c1 + self.cmu) / N / 10:
            self.eigeneval = self.counteval
            self.C = np.triu(self.C) + np.triu(self.C, 1).T
            d2, self.B = np.linalg.eigh(self.C)
            self.D = np.sqrt(np.maximum(d2, 1e-30))
            self.invsqrtC = (self.B / self.D[None, :]) @ self.B.T
        self._geno = None
        self._last_f = f[order]

    @property
    def result(self):
        xm = self.mean if self.boundary is None else self.boundary(self.mean[None, :])[0]
        return _Result((self.best_x, self.best_f, self.best_evals, self.counteval, self.countiter, xm,
                        self.sigma * np.sqrt(np.diag(self.C))))

    def stop(self):
        d = {}
        if self.maxiter is not None and self.countiter >= self.maxiter:
            d["maxiter"] = self.maxiter
        if self.sigma * float(self.D.max()) < 1e-11:
            d["tolx"] = 1e-11
        return d

    def disp(self, modulo: int = 1):
        if self.countiter == 1:
            print("Iterat #Fevals   function value  axis ratio  sigma  min&max std")
        if modulo and self.countiter % modulo == 0:
            stds = self.sigma * np.sqrt(np.diag(self.C))
            fbest_it = float(self._last_f[0]) if hasattr(self, "_last_f") else float("nan")
            print(f"{self.countiter:5d} {self.counteval:6d} {fbest_it: .15e} {self.D.max() / self.D.min():.1e} "
                  f"{self.sigma:.2e}  {stds.min():.0e}  {stds.max():.0e}")
