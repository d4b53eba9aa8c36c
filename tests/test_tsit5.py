"""CPU: Tsit5 (the method of the reference's jax path, integrator.py:56-61) in the oracle.  The tableau is pinned by its
order conditions -- every rooted tree up to order 5 for the propagated solution, up to order 4 for the embedded one -- and
the integrator by the LSODA(1e-13) answers on the reference RHS (tests/golden/tight_*.npz)."""
import ctypes as C
import itertools

import numpy as np
import pytest

import helpers as H
from oracle import oracle as O
from test_oracle_golden import TIGHT_CASES, _spec_for_integration


def _partitions(n, maxp=None):
    maxp = n if maxp is None else maxp
    if n == 0:
        yield []
        return
    for p in range(min(n, maxp), 0, -1):
        for rest in _partitions(n - p, p):
            yield [p] + rest


def _trees(n):
    if n == 1:
        return [()]
    res = set()
    for parts in _partitions(n - 1):
        for combo in itertools.product(*[_trees(p) for p in parts]):
            res.add(tuple(sorted(combo)))
    return list(res)


def _order(t):
    return 1 + sum(_order(s) for s in t)


def _gamma(t):
    g = _order(t)
    for s in t:
        g *= _gamma(s)
    return g


def test_tableau_order_conditions():
    l = O.lib()
    a, e = (C.c_double * 42)(), (C.c_double * 7)()
    l.orc_tsit5_tableau.restype = None
    l.orc_tsit5_tableau(a, e)
    A = np.zeros((7, 7))
    A[:, :6] = np.array(a[:]).reshape(7, 6)
    b, bt = A[6].copy(), np.array(e[:])
    assert np.allclose(A.sum(1), [0, 0.161, 0.327, 0.9, 0.9800255409045097, 1, 1], atol=1e-15)

    def phi(t):
        v = np.ones(7)
        for s in t:
            v = v * (A @ phi(s))
        return v

    assert sum(len(_trees(n)) for n in range(1, 6)) == 17
    for n in range(1, 6):
        for t in _trees(n):
            assert abs(b @ phi(t) - 1 / _gamma(t)) <= 3e-14, (n, t)
            if n <= 4:  # the embedded solution b - btilde: order 4 ...
                assert abs((b - bt) @ phi(t) - 1 / _gamma(t)) <= 3e-14, (n, t)
    assert max(abs((b - bt) @ phi(t) - 1 / _gamma(t)) for t in _trees(5)) > 1e-4  # ... and not 5


@pytest.mark.parametrize("fix,model", [(c[0], c[1]) for c in TIGHT_CASES])
def test_tsit5_reaches_true_solution(fix, model):
    g = H.gold("tight_" + fix)
    dt, nu = float(g["dt"]), g["u"].shape[1]
    scale = np.maximum(np.abs(g["xf"]), 1e-6 * np.max(np.abs(g["xf"]), axis=0, keepdims=True))
    res = []
    for tol in (1e-8, 1e-11):
        s = _spec_for_integration(model, dt, nu, integrator="tsit5", rtol=tol, atol=tol * 1e-2)
        xf, ns = O.integrate(s, g["x"].T, g["u"].T)
        assert np.isfinite(xf).all()
        res.append((np.max(np.abs(xf.T - g["xf"]) / scale), ns.sum(axis=0).mean()))
        sd = _spec_for_integration(model, dt, nu, integrator="dopri5", rtol=tol, atol=tol * 1e-2)
        nd = O.integrate(sd, g["x"].T, g["u"].T)[1].sum(axis=0).mean()
        assert ns.sum(axis=0).mean() <= 1.3 * nd + 2  # same class as the Dormand-Prince pair (usually a few steps fewer)
    assert res[0][0] <= 2e-5 and res[1][0] <= 5e-9, res  # igniting cstr samples amplify ~100x
    assert res[1][0] <= 0.05 * res[0][0] + 1e-11, res
