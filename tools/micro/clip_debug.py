import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import synth, barcode_trimmer, runtime
from ngspeciesid_amd._capi import ReadSet, polish_params
from oracle_lib import load_oracle
from util_seq import edit_distance
gpu = runtime.get_api(0); orc = load_oracle()
tails = barcode_trimmer.get_universal_tails()
bodies = [b.tobytes().decode() for b in synth.make_species(2, 520, 0.15, seed=8)]
amps = [tails["1_F_fw"] + b + tails["2_R_fw"] for b in bodies]
for nreads, rcf in ((40, 0.0), (40, 0.5), (1200, 0.5)):
    rd = synth.make_reads([np.frombuffer(a.encode(), dtype=np.uint8) for a in amps], nreads, mu=15.0, seed=3, rc_fraction=rcf)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32); n0 = int((spc == 0).sum())
    for bbs, nm in ((bodies, "bodies"), (amps, "amps")):
        for trim in (1, 2):
            for it in (1, 2):
                prm = polish_params(iters=it, k=13, w=20, tile_depth=6, band=0, trim=trim, aln_mode=3, stop_when_stable=0)
                a, ua = gpu.polish(ReadSet.from_strings(bbs), rs, [0, n0, rs.n], prm, read_order=order)
                b, ub = orc.polish(ReadSet.from_strings(bbs), rs, [0, n0, rs.n], prm, read_order=order)
                print(nreads, rcf, nm, "trim", trim, "iters", it, "equal", a == b, "used", ua.tolist(), ub.tolist(), "lens", [len(x) for x in a], [len(x) for x in b], "ed", [edit_distance(x, y) for x, y in zip(a, b)], flush=True)
