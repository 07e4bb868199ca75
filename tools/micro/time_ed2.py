"""dev tool: time the polisher's edit-distance aligner on what the polisher gives it: reads with an ONT-like error profile against their backbone.
    python tools/micro/time_ed2.py [lib.so] [n_pairs] [read length]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ngspeciesid_amd import runtime
args = sys.argv[1:]
if args and args[0].endswith(".so"): runtime.LIB_PATH = os.path.abspath(args.pop(0))
from ngspeciesid_amd._capi import ReadSet
api = runtime.get_api(0)
rng = np.random.default_rng(1)
n = int(args[0]) if args else 200000
L = int(args[1]) if len(args) > 1 else 750
A = np.frombuffer(b"ACGT", dtype=np.uint8)
nb = 5
backs = [A[rng.integers(0, 4, L)] for _ in range(nb)]
def mutate(s, e):
    u = rng.random(len(s)); keep = u >= 0.3 * e                       # 30 % deletions
    sub = (u >= 0.3 * e) & (u < 0.7 * e)                                # 40 % substitutions
    ins = (u >= 0.7 * e) & (u < e)                                      # 30 % insertions
    s = s.copy(); s[sub] = A[rng.integers(0, 4, int(sub.sum()))]
    rep = np.where(ins, 2, 1) * keep
    o = np.repeat(s, rep); return o
reads = []; ti = []
for i in range(2000):
    b = int(rng.integers(0, nb)); e = float(np.clip(10 ** (-rng.normal(17, 2.5) / 10) * 2.2, 0.01, 0.2))
    reads.append(mutate(backs[b], e).tobytes().decode()); ti.append(b)
Q = ReadSet.from_strings(reads); T = ReadSet.from_strings([b.tobytes().decode() for b in backs])
pick = rng.integers(0, len(reads), n)
qi = pick.astype(np.uint32); tix = np.array(ti, dtype=np.uint32)[pick]
for it in range(3):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    t = time.time(); r = api.ed_align_batch(Q, T, qi, tix, window=500, bp_windows=2); dt = time.time() - t
    buf = C.create_string_buffer(4096); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(4096))
    print("pairs", n, "wall %.3fs" % dt, buf.value.decode().strip(), "mean distance %.1f" % r[0].mean(), "checksum", int(r[0].sum()), int(r[1].sum()), int(r[2].sum()))
