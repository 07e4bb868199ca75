"""dev tool / stress parity: many independent POA groups (HIP vs the CPU oracle), to catch rare-case divergences the small tests miss.

    python tools/stress_poa.py [n_groups] [depth] [mode: spoa|polish] [seed] [band: 64|128|256]

spoa  : n_groups groups of `depth` reads -> poa_consensus (local mode, one tile per group when depth <= 8)
polish: n_groups backbones (noisy drafts) each with `depth` reads -> polish (global + semi-global window layers, aligner included)
Exit code 1 on any difference; prints the first few differing groups.  Optional env NGSID_LIB = alternative HIP library.
"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import runtime, synth
if os.environ.get("NGSID_LIB"):
    runtime.LIB_PATH = os.environ["NGSID_LIB"]
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params
from oracle_lib import load_oracle

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 500
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
what = sys.argv[3] if len(sys.argv) > 3 else "spoa"
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 11
band = int(sys.argv[5]) if len(sys.argv) > 5 else 128
api = runtime.get_api(0); orc = load_oracle()
L = int(os.environ.get("STRESS_L", 750))          # amplicon length
sp = synth.make_species(4, L, 0.15, seed=seed)
rd = synth.make_reads(sp, ng * depth, mu=17.0, seed=seed + 1, rc_fraction=0.3 if what == "polish" else 0.0)      # the polisher orients reads itself
spc = rd["species"].numpy()
order = np.argsort(spc, kind="stable").astype(np.uint32)          # groups are (mostly) single-species runs of `depth` reads
rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
grp = np.arange(0, ng * depth + 1, depth, dtype=np.uint64)
if os.environ.get("STRESS_ONLY"):            # restrict to one group (same reads as in the full run)
    g0 = int(os.environ["STRESS_ONLY"]); grp = grp[g0:g0 + 2]; ng = 1
bad = 0
if what == "spoa":
    prm = poa_params(tile_depth=8, band=band)
    t = time.time(); a = api.poa_consensus(rs, grp, prm, read_order=order); ta = time.time() - t
    t = time.time(); b = orc.poa_consensus(rs, grp, prm, read_order=order); tb = time.time() - t
    for g in range(ng):
        if a[g] != b[g]:
            bad += 1
            if bad <= 5: print("group", g, "differs: hip len", len(a[g]), "oracle len", len(b[g]))
    print("spoa groups %d depth %d: %d differ   (hip %.2fs, oracle %.2fs)" % (ng, depth, bad, ta, tb))
else:
    rng = np.random.default_rng(seed + 2)
    bbs = []
    for g in range(ng):
        s = sp[int(spc[order[g * depth]])].copy()
        m = rng.random(len(s)) < 0.03; s[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
        keep = rng.random(len(s)) >= 0.01
        bbs.append(s[keep].tobytes().decode())
    bb = ReadSet.from_strings(bbs)
    prm = polish_params(iters=2, tile_depth=8, band=band, trim=2)
    t = time.time(); a, ua = api.polish(bb, rs, grp, prm, read_order=order); ta = time.time() - t
    t = time.time(); b, ub = orc.polish(bb, rs, grp, prm, read_order=order); tb = time.time() - t
    for g in range(ng):
        if a[g] != b[g] or int(ua[g]) != int(ub[g]):
            bad += 1
            if bad <= 5: print("group", g, "differs: hip len", len(a[g]), "oracle len", len(b[g]), "used", int(ua[g]), int(ub[g]))
    print("polish groups %d depth %d: %d differ   (hip %.2fs, oracle %.2fs)" % (ng, depth, bad, ta, tb))
sys.exit(1 if bad else 0)
