"""dev tool: time the POA tile kernel alone (one group of reads; argv: n_reads [band [tile_depth]]; tile depth 8 by default, 0 = one graph in read order), optionally with an alternative .so (env NGSID_LIB)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ngspeciesid_amd import runtime, synth
if os.environ.get("NGSID_LIB"):
    runtime.LIB_PATH = os.environ["NGSID_LIB"]
from ngspeciesid_amd._capi import ReadSet, poa_params
api = runtime.get_api(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
sp = synth.make_species(1, 750, 0.15, seed=3)
rd = synth.make_reads(sp, n, mu=17.0, seed=5)
rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
for it in range(2):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    t = time.time(); c = api.poa_consensus(rs, [0, n], poa_params(tile_depth=int(sys.argv[3]) if len(sys.argv) > 3 else 8, band=int(sys.argv[2]) if len(sys.argv) > 2 else 0)); dt = time.time() - t
    buf = C.create_string_buffer(4096); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(4096))
    print(os.environ.get("NGSID_LIB", "default"), "reads", n, "wall %.3fs" % dt, buf.value.decode().strip(), "len", len(c[0]), "ok", c[0][1:-1] in sp[0].tobytes().decode() or sp[0].tobytes().decode() in c[0])
