"""dev tool / stress parity of the semi-global aligner (HIP 16-bit + 32-bit paths vs the CPU oracle) on many random pairs.

    python tools/stress_align.py [n_pairs] [max_len] [seed]

Pairs are noisy copies of random templates (substitutions, indels, lower-case letters, N wildcards, unrelated pairs, empty and
1-base sequences); lengths are drawn so that every strip-count / rows-per-lane specialisation is hit.  Exit code 1 on any difference.
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import runtime
from ngspeciesid_amd._capi import ReadSet
from oracle_lib import load_oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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


qs, ts = [], []
for i in range(n):
    kind = rng.random()
    L = int(rng.integers(1, maxlen + 1)) if kind > 0.05 else int(rng.integers(0, 3))
    base = A[rng.integers(0, 4, L)]
    q = mutate(base, 0.08, 0.08); t = mutate(base, 0.05, 0.05) if kind > 0.15 else A[rng.integers(0, 4, int(rng.integers(0, maxlen + 1)))]
    for s in (q, t):
        if len(s) and rng.random() < 0.2: s[rng.integers(0, len(s), max(1, len(s) // 50))] = ord("N")
        if len(s) and rng.random() < 0.1:
            m = rng.random(len(s)) < 0.3; s[m] = s[m] | 0x20                     # lower case
    qs.append(q.tobytes().decode()); ts.append(t.tobytes().decode())
q = ReadSet.from_strings(qs); t = ReadSet.from_strings(ts)
idx = np.arange(n, dtype=np.uint32)
opens = rng.integers(2, 6, n).astype(np.int32); mids = rng.integers(-1, 14, n).astype(np.int32)
bad = 0
import ctypes as C
for env, tag in ((None, "16-bit dispatch"), ("1", "32-bit only")):
    assert api.lib.ngsid_ctx_option(api.ctx, b"align32", C.c_int64(1 if env else 0)) == 0
    t0 = time.time(); got = api.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids); tg = time.time() - t0
    if tag.startswith("16"):
        t0 = time.time(); exp = orc.sg_align_batch(q, t, idx, idx, opens, 1, 2, -2, 13, mids); to = time.time() - t0
    for nm, a, b in zip(["score", "ncols", "nmatch", "region"], got, exp):
        d = np.nonzero(a != b)[0]
        if len(d):
            bad += len(d); print(tag, nm, "differs at", d[:6], "got", a[d[:6]], "exp", b[d[:6]], "lens", [(len(qs[i]), len(ts[i])) for i in d[:6]])
    print("%s: %d pairs, max len %d   (hip %.2fs, oracle %.2fs)" % (tag, n, maxlen, tg, to))
sys.exit(1 if bad else 0)
