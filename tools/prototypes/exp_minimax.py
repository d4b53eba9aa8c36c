#!/usr/bin/env python3
"""Remez fit of exp(r) on |r| <= ln2/2 (+ margin) in the form 1 + r + r^2 q(r), relative error, for exp_bounded
(pcg_pack.hpp).  Prints the double-rounded coefficients and the achieved error with them.  usage: exp_minimax.py [degree]"""
import sys

import mpmath as mp

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 10
A = mp.log(2) / 2 * (1 + mp.mpf(2) ** -20)  # rounding of n = rint(x log2 e) can leave |r| a hair above ln2/2
NQ = DEG - 1  # free coefficients c2..cDEG


def relerr(c, r):
    p = 1 + r + sum(ck * r ** (k + 2) for k, ck in enumerate(c))
    return p / mp.exp(r) - 1


def remez():
    n = NQ + 1
    xs = [-A * mp.cos(mp.pi * i / (n - 1 + 1)) for i in range(n + 0)]
    xs = [A * mp.cos(mp.pi * (n - i) / n) for i in range(n + 1)][: n + 0]
    # n+? points: unknowns NQ coefficients + E -> NQ + 1 points
    xs = [A * mp.cos(mp.pi * (NQ - i) / NQ) for i in range(NQ + 1)]
    for it in range(40):
        M = mp.matrix(NQ + 1, NQ + 1)
        b = mp.matrix(NQ + 1, 1)
        for i, x in enumerate(xs):
            ex = mp.exp(x)
            for k in range(NQ):
                M[i, k] = x ** (k + 2) / ex
            M[i, NQ] = (-1) ** i
            b[i] = 1 - (1 + x) / ex
        sol = mp.lu_solve(M, b)
        c = [sol[k] for k in range(NQ)]
        E = sol[NQ]
        # new extrema: scan
        N = 4000
        grid = [-A + 2 * A * j / N for j in range(N + 1)]
        vals = [relerr(c, g) for g in grid]
        ext = []
        for j in range(N + 1):
            l = vals[j - 1] if j > 0 else None
            r = vals[j + 1] if j < N else None
            v = vals[j]
            if (l is None or abs(v) >= abs(l)) and (r is None or abs(v) >= abs(r)):
                if ext and (v > 0) == (ext[-1][1] > 0):
                    if abs(v) > abs(ext[-1][1]):
                        ext[-1] = (grid[j], v)
                else:
                    ext.append((grid[j], v))
        if len(ext) < NQ + 1:
            break
        # keep the NQ+1 largest consecutive alternating
        while len(ext) > NQ + 1:
            if abs(ext[0][1]) < abs(ext[-1][1]):
                ext.pop(0)
            else:
                ext.pop()
        new = [e[0] for e in ext]
        mx = max(abs(e[1]) for e in ext)
        if abs(mx - abs(E)) < abs(E) * mp.mpf(10) ** -6:
            xs = new
            break
        xs = new
    return c, E


c, E = remez()
print("degree", DEG, "levelled relative error", mp.nstr(abs(E), 5))
cd = [float(x) for x in c]
worst = max(abs(relerr([mp.mpf(v) for v in cd], -A + 2 * A * j / 20000)) for j in range(20001))
print("with double-rounded coefficients: max relative error", mp.nstr(worst, 5))
for k, v in enumerate(cd):
    print(f"  c{k + 2} = {v!r}   ({v.hex()})")
