"""dev tool / stress parity of the edit-distance polisher aligner (HIP bit-parallel kernel vs the oracle's plain DP).

    python tools/stress_ed.py [n_pairs] [max_qlen] [seed] [error_scale]
(error_scale multiplies the 8 % substitution / 8 % indel rates of the related queries: small values give long queries with a small distance,
the case the banded sliding-window instance of the kernel is for)
Exit code 1 on any difference in distance, span or window break points.
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import runtime
from ngspeciesid_amd._capi import ReadSet
from oracle_lib import load_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
maxq = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
escale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
full = len(sys.argv) > 5 and sys.argv[5] == "full"      # full-length queries against targets of similar length (what the polisher sees): whole waves stay inside the band
rng = np.random.default_rng(seed)
api = runtime.get_api(0); orc = load_oracle()
A = np.frombuffer(b"ACGT", dtype=np.uint8)


def mutate(s, sub, indel):
    out = []
    for ch in s:
        u = rng.random()
        if u < indel / 2: continue
        if u < indel: out.append(A[rng.integers(0, 4)])
        out.append(A[rng.integers(0, 4)] if rng.random() < sub else ch)
    return np.array(out, dtype=np.uint8)


nt = 40
targets = [A[rng.integers(0, 4, int(rng.integers(int(0.93 * maxq), maxq) if full else rng.integers(1, maxq + 200)))] for _ in range(nt)]
if not full: targets[0] = A[rng.integers(0, 4, 1)]
qs, ti = [], []
for i in range(n):
    k = int(rng.integers(0, nt)); T = targets[k]; kind = rng.random()
    if full and i < 0.9 * n: kind = 1.0
    if kind < 0.04: q = A[rng.integers(0, 4, int(rng.integers(0, 3)))]                      # empty / tiny
    elif kind < 0.12: q = A[rng.integers(0, 4, int(rng.integers(1, maxq + 1)))]             # unrelated
    else:
        a = int(rng.integers(0, len(T))); b = int(rng.integers(a, min(len(T), a + maxq) + 1))
        if full: a, b = int(rng.integers(0, 8)), len(T) - int(rng.integers(0, 8))
        q = mutate(T[a:b], 0.08 * escale, 0.08 * escale)[:maxq]
    if len(q) and rng.random() < 0.2: q[rng.integers(0, len(q), max(1, len(q) // 40))] = ord("N")
    if len(q) and rng.random() < 0.1:
        m = rng.random(len(q)) < 0.3; q[m] = q[m] | 0x20
    qs.append(q.tobytes().decode()); ti.append(k)
tstr = []
for T in targets:
    T = T.copy()
    if rng.random() < 0.3 and len(T): T[rng.integers(0, len(T), max(1, len(T) // 60))] = ord("N")
    tstr.append(T.tobytes().decode())
Q = ReadSet.from_strings(qs); T = ReadSet.from_strings(tstr)
qi = np.arange(n, dtype=np.uint32); ti = np.array(ti, dtype=np.uint32)
W = 100; nw = (max(len(t) for t in tstr) + W - 1) // W
t0 = time.time(); g = api.ed_align_batch(Q, T, qi, ti, window=W, bp_windows=nw); tg = time.time() - t0
t0 = time.time(); e = orc.ed_align_batch(Q, T, qi, ti, window=W, bp_windows=nw); to = time.time() - t0
bad = 0
for nm, a, b in zip(["distance", "span", "bp"], g, e):
    d = np.nonzero((a != b).reshape(n, -1).any(axis=1))[0]
    if len(d):
        bad += len(d); x = int(d[0])
        print(nm, "differs for", len(d), "pairs, first", x, "qlen", len(qs[x]), "tlen", len(tstr[int(ti[x])]), "got", a[x].ravel()[:12], "exp", b[x].ravel()[:12])
print("edit-distance aligner: %d pairs, max query %d: %d differences   (hip %.2fs, oracle %.2fs)" % (n, maxq, bad, tg, to))
sys.exit(1 if bad else 0)
