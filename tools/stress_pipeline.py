"""volume parity of the WHOLE hot path: random small read sets (species count, length, depth, error profile, strand mix, k/w, tile depth, band,
iterations) through pipeline.run_hot_path on the HIP library and on the CPU oracle: cluster map, counters and every draft / polished sequence
must be identical.    python tools/stress_pipeline.py n_trials seed"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import load_oracle
from ngspeciesid_amd import runtime, synth, pipeline
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd.hostutil import subset_reads
ntr = int(sys.argv[1]) if len(sys.argv) > 1 else 60; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
api = runtime.get_api(0); orc = load_oracle()
rng = np.random.default_rng(seed)
KW = [(13, 20), (15, 50), (25, 30), (13, 18), (10, 20), (30, 35)]
bad = 0; t0 = time.time()
for trial in range(ntr):
    k, w = KW[int(rng.integers(0, len(KW)))]
    pt = select_p_table(k, w)
    if np.isnan(pt).all(): continue
    nsp = int(rng.integers(1, 4)); L = int(rng.choice([90, 180, 420, 507, 760, 1003, 1300])); n = int(rng.integers(30, 260)); mu = float(rng.choice([12.0, 14.0, 17.0, 25.0]))
    rcf = float(rng.choice([0.0, 0.0, 0.5])); D = int(rng.choice([0, 3, 4, 4, 6, 8])); band = int(rng.choice([0, 64, 128])); iters = int(rng.integers(0, 4))
    sp = synth.make_species(nsp, L, float(rng.choice([0.1, 0.2])), seed=int(rng.integers(1, 1 << 30)))
    rd = synth.make_reads(sp, n, mu=mu, seed=int(rng.integers(1, 1 << 30)), rc_fraction=rcf)
    rs0 = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    score, err, keep = api.score_reads(rs0, k, 7.0)
    s2, e2, k2 = orc.score_reads(rs0, k, 7.0)
    if not (np.array_equal(score, s2) and np.array_equal(keep, k2)): bad += 1; print("trial", trial, "score differs"); continue
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    if len(idx) < 4: continue
    sub = subset_reads(rs0, idx)
    kw = dict(acc_rank=np.arange(sub.n, dtype=np.uint32), k=k, w=w, abundance_ratio=float(rng.choice([0.02, 0.1])), racon_iter=iters, tile_depth=D, band=band, p_shared=pt,
              polish_stop_when_stable=bool(rng.integers(0, 2)), do_polish=iters > 0)
    try:
        a = pipeline.run_hot_path(api, sub, score[idx], **kw); b = pipeline.run_hot_path(orc, sub, score[idx], **kw)
    except Exception as ex:
        bad += 1; print("trial", trial, (k, w, nsp, L, n, mu, rcf, D, band, iters), "raised", repr(ex)[:300]); continue
    ok = np.array_equal(a["rep_of"], b["rep_of"]) and np.array_equal(a["counters"], b["counters"]) and [(c[0], c[1], c[2], c[3]) for c in a["centers"]] == [(c[0], c[1], c[2], c[3]) for c in b["centers"]]
    if not ok:
        bad += 1; print("trial", trial, (k, w, nsp, L, n, mu, rcf, D, band, iters), "DIFFERS: clusters equal", np.array_equal(a["rep_of"], b["rep_of"]), "centres", len(a["centers"]), len(b["centers"]))
print("%d trials, %d bad, %.0f s" % (ntr, bad, time.time() - t0))
sys.exit(1 if bad else 0)
