"""dev tool: k_score_reads on 1 M synthetic 750 bp reads (HIP-event time) and a checksum of the scores."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp = synth.make_species(5, 750, 0.15, seed=1)
rd = synth.make_reads(sp, n, mu=17.0, seed=2, device=dev) if "device" in synth.make_reads.__code__.co_varnames else synth.make_reads(sp, n, mu=17.0, seed=2)
rs = ReadSet.from_torch(rd["seq"].to(dev), rd["qual"].to(dev), rd["off"].to(dev))
for rep in range(3):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    score, err, keep = api.score_reads(rs, 13, 7.0)
    buf = C.create_string_buffer(1 << 12); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(len(buf)))
    print(buf.value.decode().strip(), "kept", int(keep.sum()), "score sum %.6f" % float(score.sum()), "err sum %.9f" % float(err.sum()), flush=True)
