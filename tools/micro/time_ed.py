"""dev tool: time the edit-distance aligner kernel alone (pairs of ~750 bp)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ngspeciesid_amd import runtime
from ngspeciesid_amd._capi import ReadSet
api = runtime.get_api(0)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
L = 750
base = rng.integers(0, 4, L)
seqs = []
for i in range(64):
    b = base.copy(); m = rng.random(L) < 0.05; b[m] = rng.integers(0, 4, int(m.sum())); seqs.append("".join("ACGT"[x] for x in b))
q = ReadSet.from_strings(seqs); qi = rng.integers(0, 64, n).astype(np.uint32); ti = rng.integers(0, 64, n).astype(np.uint32)
for it in range(3):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    t = time.time(); r = api.ed_align_batch(q, q, qi, ti, window=500, bp_windows=2); dt = time.time() - t
    buf = C.create_string_buffer(4096); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(4096))
    print("pairs", n, "wall %.3fs" % dt, buf.value.decode().strip(), "checksum", int(r[0].sum()), int(r[1].sum()))
