"""dev tool: time the aligner kernel alone (pairs of ~750 bp), optionally with an alternative .so (env NGSID_LIB)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ngspeciesid_amd import runtime
if os.environ.get("NGSID_LIB"):
    runtime.LIB_PATH = os.environ["NGSID_LIB"]
from ngspeciesid_amd._capi import ReadSet
api = runtime.get_api(0)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = 750
base = rng.integers(0, 4, L)
seqs = []
for i in range(64):
    b = base.copy(); m = rng.random(L) < 0.05; b[m] = rng.integers(0, 4, int(m.sum())); seqs.append("".join("ACGT"[x] for x in b))
q = ReadSet.from_strings(seqs); qi = rng.integers(0, 64, n).astype(np.uint32); ti = rng.integers(0, 64, n).astype(np.uint32)
for it in range(3):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    t = time.time(); r = api.sg_align_batch(q, q, qi, ti, 3, 1, 2, -2, 13, None); dt = time.time() - t
    buf = C.create_string_buffer(4096); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(4096))
    print(os.environ.get("NGSID_LIB", "default"), "pairs", n, "wall %.3fs" % dt, buf.value.decode().strip(), "checksum", int(r[0].sum()), int(r[3].sum()))
