"""Rodas5 (8 stages, order 5(4)) against Rodas4 (6 stages, 4(3)) on the 10-state extraction cascade, CPU oracle, the action
box of BASELINE configs[2] (tests/test_rodas4.py::_me_box): attempts per env step and worst relative error of one env step
against a 1e-13 solve, per tolerance.  Cost model: an attempt of the fifth-order pair is 8 (RHS + solve) against 6."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import oracle as O
from test_rodas4 import _me_box

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
spec, cases, refs = _me_box(B, 5)
l = O.lib()



def run(order, tol, epc=True, ctrl=None):
    kw = {} if epc is True else {"endpoint_control": epc}
    sp = spec(integrator="rodas%d" % order, rtol=tol, atol=tol, cooperative=False, **kw)
    st, er = [], []
    for (xx, uu), ref in zip(cases, refs):
        y, ns = O.integrate(sp, xx, uu)
        st.append(ns.sum(0))
        er.append(np.max(np.abs(y - ref) / np.abs(ref), axis=0))
    return np.concatenate(st), np.concatenate(er)


if __name__ == "__main__":
    for order, tols in ((4, (3e-8,)), (5, (3e-8, 1e-7, 2e-7, 3e-7, 5e-7, 1e-6))):
        for tol in tols:
            for epc in (True, False):
                st, er = run(order, tol, epc)
                print(f"rodas{order} tol {tol:7.1e} ep {str(epc):5s} attempts mean {st.mean():6.2f} p99 {np.percentile(st,99):5.0f} max {st.max():4d} "
                      f"worst err {er.max():.2e} p99.9 {np.percentile(er,99.9):.2e}  cost(x stages) {st.mean()*(6 if order==4 else 8):7.1f}", flush=True)
