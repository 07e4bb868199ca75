"""dev tool: the banded / windowed edit-distance launches against the unbanded ones on the GPU (no oracle: sizes the oracle cannot do),
large batches of full-length reads, the polisher's situation.   python tools/micro/check_ed_band.py [n_pairs] [length] [error] [seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ngspeciesid_amd import runtime
from ngspeciesid_amd._capi import ReadSet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
err = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(seed); A = np.frombuffer(b"ACGT", dtype=np.uint8)
api = runtime.get_api(0)
T = [A[rng.integers(0, 4, L + int(rng.integers(-20, 20)))] for _ in range(5)]
qs, ti = [], []
for i in range(n):
    k = int(rng.integers(0, 5)); t = T[k]; e = err * float(rng.choice([0.1, 0.3, 0.6, 1.0, 1.5]))
    u = rng.random(len(t)); keep = u >= e / 3
    q = t.copy(); sub = (u >= e / 3) & (u < 2 * e / 3); q[sub] = A[rng.integers(0, 4, int(sub.sum()))]
    q = q[keep]
    ins = np.nonzero(rng.random(len(q)) < e / 3)[0]
    if len(ins): q = np.insert(q, ins, A[rng.integers(0, 4, len(ins))])
    qs.append(q.tobytes().decode()); ti.append(k)
Q = ReadSet.from_strings(qs); TT = ReadSet.from_strings([t.tobytes().decode() for t in T])
qi = np.arange(n, dtype=np.uint32); ti = np.array(ti, dtype=np.uint32); W = 500; nw = (L + 20 + W - 1) // W
res = {}
import ctypes as C
for band in ("", "0"):
    assert api.lib.ngsid_ctx_option(api.ctx, b"ed_band", C.c_int64(int(band) if band else -1)) == 0
    for rep in range(2):
        t0 = time.time(); res[band] = api.ed_align_batch(Q, TT, qi, ti, window=W, bp_windows=nw); dt = time.time() - t0
    print("band %-7s %.3fs  mean distance %.1f" % (band or "default", dt, float(res[band][0].mean())), flush=True)
bad = 0
for nm, a, b in zip(["distance", "span", "bp"], res[""], res["0"]):
    d = np.nonzero((a != b).reshape(n, -1).any(axis=1))[0]
    if len(d): bad += len(d); print(nm, "differs for", len(d), "pairs, first", d[:8], "dist", res["0"][0][d[:8]])
print("pairs %d length %d: %d differences" % (n, L, bad))
sys.exit(1 if bad else 0)
