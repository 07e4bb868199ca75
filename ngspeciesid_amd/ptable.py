"""The empirical shared-minimizer table (data; modules/p_minimizers_shared.py:1-3) and its selection (NGSpeciesID:72-77)."""
import os
import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "p_minimizers_shared.npz")
_cache = None


def read_empirical_p():
    """Rows (k, w, p, e1, e2) in the reference's order, bit-exact doubles."""
    global _cache
    if _cache is None:
        z = np.load(_DATA)
        _cache = (z["k"].astype(np.int64), z["w"].astype(np.int64), z["p"], z["e1"].astype(np.int64), z["e2"].astype(np.int64))
    return _cache


def select_p_table(k: int, w: int):
    """225 doubles [(i-1)*15+(j-1)] for e1=i/100, e2=j/100; NaN where the reference dict has no key.
    Rows with k==args.k and abs(w-args.w)<=2, symmetrised, later rows win (NGSpeciesID:74-77)."""
    kk, ww, p, e1, e2 = read_empirical_p()
    t = np.full(225, np.nan)
    m = (kk == k) & (np.abs(ww - w) <= 2)
    for pi, a, b in zip(p[m], e1[m], e2[m]):
        t[(a - 1) * 15 + (b - 1)] = pi
        t[(b - 1) * 15 + (a - 1)] = pi
    return t


def p_emp_probs_dict(k: int, w: int):
    """The dict {(e1,e2): p} exactly as NGSpeciesID:73-77 builds it."""
    t = select_p_table(k, w)
    d = {}
    for i in range(1, 16):
        for j in range(1, 16):
            v = t[(i - 1) * 15 + (j - 1)]
            if not np.isnan(v):
                d[(i / 100.0, j / 100.0)] = float(v)
    return d


def dict_to_table(p_emp_probs):
    t = np.full(225, np.nan)
    for (a, b), v in p_emp_probs.items():
        i, j = int(round(a * 100)), int(round(b * 100))
        if 1 <= i <= 15 and 1 <= j <= 15:
            t[(i - 1) * 15 + (j - 1)] = v
    return t
